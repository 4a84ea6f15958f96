P=mmlrec-a-unified-multi-task-and-multi-scenario-learning-benchmark-for-recommendation_amd
bash tools/lab/panel_pmc.sh product 2>&1 | grep -E "SQ_BUSY_CU_CYCLES|SQ_VALU_MFMA_BUSY|SQ_INSTS_MFMA|SQ_INSTS_VALU|SQ_WAVE_CYCLES"
bash tools/lab/panel_pmc.sh mfmaonly MMLREC_LIB=$PWD/$P/lib/lab/libmmlrec_panel_lab16407.so 2>&1 | grep -E "SQ_BUSY_CU_CYCLES|SQ_VALU_MFMA_BUSY|SQ_INSTS_MFMA|SQ_INSTS_VALU|SQ_WAVE_CYCLES"
GEMM_MASK=1 GEMM_AMAX_OUT=1 FWD_ONLY=1 CASES='L1 experts+gates' python3 tools/bench_gemm.py | tail -1
MMLREC_LIB=$PWD/$P/lib/lab/libmmlrec_panel_lab16407.so GEMM_MASK=1 GEMM_AMAX_OUT=1 FWD_ONLY=1 CASES='L1 experts+gates' python3 tools/bench_gemm.py | tail -1
