O=gpurun_out/ub11; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_adjoint.py tests/test_gpu_configs.py -m gpu -x -q --durations=8 > $O/new_tests.log 2>&1; echo "new tests rc $?"; tail -15 $O/new_tests.log
timeout 2400 python tools/variants.py run full_break python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_codegen_matrix.py > $O/full_break_suite.log 2>&1; echo "BREAK=2 suite rc $?"; tail -4 $O/full_break_suite.log
python tools/variants.py restore
