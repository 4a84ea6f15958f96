bash tools/lab/trace_step.sh gpurun_out/ae30_pl --no-configs --serial > gpurun_out/ae30_pl.txt 2>&1
echo WS; grep "gemm_ws\|window" gpurun_out/ae30_pl.txt | cut -c1-110
