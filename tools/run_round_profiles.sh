cd /root/repo
bash tools/profile.sh > gpurun_out/profile_sh.log 2>&1
timeout 300 bash tools/profile_cmd.sh sphC python tools/bench_scene.py sphere > /dev/null 2>&1
timeout 100 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-roofline > gpurun_out/bench_c4.json 2> /dev/null
timeout 300 python tools/config5.py --res 1024 --spp 64 --steps 2 > gpurun_out/c5_full.log 2>&1
timeout 200 python tools/tutorial_timing.py > gpurun_out/tutorial_timing.log 2>&1
timeout 200 python tools/opt_step_timing.py > gpurun_out/opt_step.log 2>&1
timeout 100 python tools/time_bwd.py > gpurun_out/time_bwd.log 2>&1
for f in gpurun_out/bench_c4.json gpurun_out/c5_full.log gpurun_out/tutorial_timing.log gpurun_out/opt_step.log gpurun_out/time_bwd.log; do tail -n 3 $f | cut -c1-300; done
timeout 200 python tools/time_bwd_env.py > gpurun_out/time_bwd_env.log 2>&1
timeout 100 python tools/env_mass_timing.py > gpurun_out/env_mass.log 2>&1
for s in microfacet conductor sphere; do timeout 100 python tools/bench_scene.py $s | tail -1; done > gpurun_out/bench_scenes.log 2>&1
timeout 300 bash tools/profile_cmd.sh c5C python tools/config5.py --res 512 --spp 16 --steps 2 > /dev/null 2>&1
timeout 200 python tools/opt_step_env_timing.py > gpurun_out/opt_step_env.log 2>&1
for f in gpurun_out/time_bwd_env.log gpurun_out/env_mass.log gpurun_out/bench_scenes.log gpurun_out/opt_step_env.log; do tail -n 4 $f | cut -c1-300; done
timeout 200 python tools/time_bwd_mat.py > gpurun_out/time_bwd_mat.log 2>&1; grep -v amdgpu gpurun_out/time_bwd_mat.log | tail -n 16
