mkdir -p gpurun_out/r02o
timeout 900 python -m pytest tests/test_gpu_plane_ops.py tests/test_gpu_me.py -x -q 2>&1 | tail -5
timeout 600 python tools/bench_me.py --tile-only --cpu > gpurun_out/r02o/me_graph.jsonl 2> gpurun_out/r02o/me_graph.err; tail -2 gpurun_out/r02o/me_graph.err
R1_ME_NO_GRAPH=1 timeout 600 python tools/bench_me.py --tile-only > gpurun_out/r02o/me_nograph.jsonl 2>/dev/null
cut -c1-330 gpurun_out/r02o/me_graph.jsonl gpurun_out/r02o/me_nograph.jsonl
timeout 300 python tools/frame_pipeline.py 2>&1 | grep "^{" | cut -c1-600
