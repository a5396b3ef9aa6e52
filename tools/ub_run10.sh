O=gpurun_out/ub10; mkdir -p $O
bash tools/ub_run9.sh nosteal_raw m_nosteal 2>&1 | tee $O/nosteal.txt
timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -30 $O/pytest_gpu.log
