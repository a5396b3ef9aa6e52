for v in brk_sb def4; do top=$(python tools/variants.py stage $v); echo "=== $v"; python tools/tri_rows.py --pkg $top 2>&1 | grep -v amdgpu.ids | head -120; done
