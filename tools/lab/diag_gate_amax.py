import sys, os, torch, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mmlrec_amd
from mmlrec_amd import ops, _lib as L
dev = torch.device("cuda:0")
for B, H, Gd in ((64, 16, 16), (4096, 128, 64)):
    E = [torch.randn(B, H, device=dev) for _ in range(4)]
    gates = [dict(G=torch.randn(B, Gd, device=dev), Wg=torch.randn(4, Gd, device=dev), P=torch.empty(B, 4, device=dev),
                  mix=torch.empty(B, H, device=dev), expert=[0, 1, 2, 3]) for _ in range(2)]
    grp = ops.make_gate_group(E, gates, B, H)
    slots = ops.amax_slots(3, dev)
    grp.amax_mix = slots[0].data_ptr()
    print("sizeof", C.sizeof(grp), "amax_mix", hex(grp.amax_mix))
    ops.gate_mix_fwd(grp)
    torch.cuda.synchronize()
    print(B, H, "slot", ops.amax_value(slots[0]), "true", max(float(g["mix"].abs().max()) for g in gates), slots[0].tolist())
