import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
full = len(sys.argv) > 1
print(d["ms_per_step"], {k[:48]: v for k, v in d["kernels_ms_per_step"].items() if full or "gemm" in k})
