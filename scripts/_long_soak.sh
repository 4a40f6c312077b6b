mkdir -p gpurun_out
timeout 1500 python tests/soak/soak_parity.py 300 > gpurun_out/soak_parity300.txt 2>&1; tail -1 gpurun_out/soak_parity300.txt
timeout 900 python tests/soak/soak_grid64.py 120 > gpurun_out/soak_grid120.txt 2>&1; tail -1 gpurun_out/soak_grid120.txt
timeout 600 python tests/soak/soak_pipeline.py 6 > gpurun_out/soak_pipe6.txt 2>&1; tail -1 gpurun_out/soak_pipe6.txt
