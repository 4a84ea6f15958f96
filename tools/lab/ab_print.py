import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], {k[:45]:v for k,v in d["kernels_ms_per_step"].items() if "gemm" in k})
