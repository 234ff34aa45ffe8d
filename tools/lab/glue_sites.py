"""Where the torch-native elementwise / copy ops of one supervised iteration are issued from: every aten op that is not a GEMM,
with the bytes it moves, grouped by the innermost python frame inside geot_amd/ (TorchDispatchMode + traceback; the backward's
ops are attributed to the autograd Function / module that created them when they run inside a custom backward, else to "autograd").
usage: glue_sites.py [clouds]"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geot_amd
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd import train_step as ts
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV)
step = ts.SupervisedStep(m)
b = _sup_batches(B, 24000)[0]
pre = step.lookahead_work(b[0])
step.iteration(b[0], b[1], b[2], pre, None)
sites = collections.defaultdict(lambda: [0, 0, collections.Counter()])
SKIP = ("mm", "bmm", "addmm", "baddbmm", "view", "_unsafe_view", "t", "transpose", "reshape", "expand", "detach", "alias", "permute",
        "unsqueeze", "squeeze", "select", "slice", "as_strided", "empty", "empty_like", "empty_strided", "split_with_sizes", "unbind",
        "_local_scalar_dense", "is_same_size", "stride", "size", "sym_size", "lift_fresh", "unsafe_split", "split", "chunk")


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name in SKIP:
            return out
        nbytes = 0
        for t in list(args) + ([out] if torch.is_tensor(out) else list(out) if isinstance(out, (tuple, list)) else []):
            if torch.is_tensor(t) and t.is_cuda:
                nbytes += t.numel() * t.element_size()
        if nbytes == 0:
            return out
        frame = "autograd / torch internals"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "geot_amd/" in fr.filename and "tools/" not in fr.filename:
                frame = "%s:%d %s" % (fr.filename.split("geot_amd/")[-1], fr.lineno, fr.name)
                break
        s = sites[frame]
        s[0] += 1; s[1] += nbytes; s[2][name] += 1
        return out


with Sites():
    step.iteration(b[0], b[1], b[2], pre, None)
torch.cuda.synchronize()
tot = sum(v[1] for v in sites.values())
print("%d non-GEMM aten calls, %.2f GB touched (at 4 TB/s: %.2f ms + ~5 us per launch)" % (sum(v[0] for v in sites.values()), tot / 1e9, tot / 4e9))
for k, v in sorted(sites.items(), key=lambda kv: -(kv[1][1] / 4e6 + 5 * kv[1][0]))[:45]:
    print("%5d calls %8.1f MB  ~%6.0f us  %-70s %s" % (v[0], v[1] / 1e6, v[1] / 4e6 + 5 * v[0], k[:70], dict(v[2].most_common(4))))
