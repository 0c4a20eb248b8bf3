set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { tail -40 gpurun_out/t_gpu.log; exit 1; }
tail -1 gpurun_out/t_gpu.log
for a in "dragon 1" "dragon 8" "bunny 1" "bunny 8"; do
timeout -k 10 120 python3 tools/round_profile.py $a 0 | sed -n '1p;4p' | cut -c1-20,62-330 | sed "s/^/$a: /"
done
