P=mmlrec-a-unified-multi-task-and-multi-scenario-learning-benchmark-for-recommendation_amd
MMLREC_LIB=$PWD/$P/lib/lab/libmmlrec_ws_nt.so bash tools/lab/trace_step.sh gpurun_out/ae30_nt --no-configs --serial > gpurun_out/ae30_nt.txt 2>&1
bash tools/lab/trace_step.sh gpurun_out/ae30_pl --no-configs --serial > gpurun_out/ae30_pl.txt 2>&1
echo NT; grep "gemm_ws\|window" gpurun_out/ae30_nt.txt | cut -c1-110
echo PLAIN; grep "gemm_ws\|window" gpurun_out/ae30_pl.txt | cut -c1-110
