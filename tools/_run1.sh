set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { tail -40 gpurun_out/t_gpu.log; exit 1; }
tail -1 gpurun_out/t_gpu.log
timeout -k 10 120 python3 __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 400 python3 bench.py > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err || { tail -20 gpurun_out/bench_h.err; exit 1; }
python3 tools/bench_brief.py gpurun_out/bench_h.json
for st in 2 4 8; do
timeout -k 10 200 python3 bench.py --no-cpu-baseline --pixel-stride $st > gpurun_out/bench_h$st.json 2> gpurun_out/bench_h$st.err
python3 tools/bench_brief.py gpurun_out/bench_h$st.json | head -1
done
timeout -k 10 200 python3 bench.py --scene bunny > gpurun_out/bench_hb.json 2> gpurun_out/bench_hb.err
python3 tools/bench_brief.py gpurun_out/bench_hb.json
timeout -k 10 200 python3 bench.py --no-cpu-baseline --scene bunny --pixel-stride 8 > gpurun_out/bench_hb8.json 2> gpurun_out/bench_hb8.err
python3 tools/bench_brief.py gpurun_out/bench_hb8.json | head -1
python3 tools/make_standin_scene.py lucy > /dev/null
timeout -k 10 500 python3 bench.py --scene scenes/_generated/CBlucy_standin.dae --width 1920 --height 1080 --spp 4096 --depth 8 --steps 2 --warmup 1 --cpu-spp 64 > gpurun_out/bench_hl.json 2> gpurun_out/bench_hl.err
python3 tools/bench_brief.py gpurun_out/bench_hl.json | head -3
timeout -k 10 300 python3 bench.py --scene spheres --depth 16 --no-cpu-baseline > gpurun_out/bench_hs.json 2> gpurun_out/bench_hs.err
python3 tools/bench_brief.py gpurun_out/bench_hs.json | head -1
timeout -k 10 300 python3 bench.py --scene spheres --depth 16 --specular --no-cpu-baseline > gpurun_out/bench_hsX.json 2> gpurun_out/bench_hsX.err
python3 tools/bench_brief.py gpurun_out/bench_hsX.json | head -1
for st in 2 4 8; do
timeout -k 10 200 python3 bench.py --no-cpu-baseline --pixel-stride $st --per-pixel 1 > gpurun_out/bench_hp$st.json 2> gpurun_out/bench_hp$st.err
python3 tools/bench_brief.py gpurun_out/bench_hp$st.json | head -1 | sed 's/^/per-pixel: /'
done
