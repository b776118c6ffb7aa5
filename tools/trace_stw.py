#!/usr/bin/env python3
"""Developer tool (make -C vp-suite_amd/csrc ablate; VPX_LIB=build/libvpx_ablate.so): every workgroup of ONE stw_kernel launch (the deferred weight
gradients of a pass: vpx_stlstm_wgrad_batch over IMGS images of 16x16) — which XCD / CU it ran on, its pass (0, 1 long; 2 short; 3 centre tap),
start and end of its item loop -> per XCD: workgroups by pass, busy time of its CUs, when its last workgroup finished. IMGS, CH, VPX_STW_NS."""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VPX_LIB", os.path.join(ROOT, "build", "libvpx_ablate.so"))
sys.path.insert(0, ROOT)
import torch
import vp_suite_amd as v
L = v._lib.lib()
imgs = int(os.environ.get("IMGS", 4864)); Ch = int(os.environ.get("CH", 128)); Cin = int(os.environ.get("CIN", Ch)); H = W = 16
from vp_suite_amd._lib import STLSTMDesc
desc = STLSTMDesc(imgs, Cin, Ch, H, W, 5, 0, v._lib.LAYOUT_NHWC, v.ops.PRECISIONS["bf16x3"], v._lib.FLAG_SAVE_FOR_BWD)
L.vpx_stlstm_wgrad_batch_workspace_bytes.restype = ctypes.c_size_t
need = L.vpx_stlstm_wgrad_batch_workspace_bytes(ctypes.byref(desc))
ws = torch.empty(need, dtype=torch.uint8, device="cuda")
HW = H * W
dg8 = torch.zeros(imgs * HW * 8 * Ch * 4, dtype=torch.uint8, device="cuda")
srcs = [torch.zeros(imgs * HW * c * 4, dtype=torch.uint8, device="cuda") for c in (Cin, Ch, Ch, Ch, Ch)]
k = 5
dW = [torch.empty(s, device="cuda") for s in ((7 * Ch, Cin, k, k), (4 * Ch, Ch, k, k), (3 * Ch, Ch, k, k), (Ch, 2 * Ch, k, k), (Ch, 2 * Ch, 1, 1))]
arr = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in srcs])
for _ in range(3):
    rc = L.vpx_stlstm_wgrad_batch(ctypes.byref(desc), ctypes.c_void_p(dg8.data_ptr()), arr, *[ctypes.c_void_p(t.data_ptr()) for t in dW],
                                  ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(need), None)
    assert rc == 0, L.vpx_last_error()
torch.cuda.synchronize()
n = 8192
buf = (ctypes.c_ulonglong * (n * 4))()
L.vpx_dbg_stw_trace.argtypes = [ctypes.c_void_p]
assert L.vpx_dbg_stw_trace(buf) == 0
rows = [(i, buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n) if buf[4 * i + 1] > buf[4 * i] > 0]
t0 = 0   # (s_memtime bases differ between XCDs: spans are taken per XCD below)
print(f"{len(rows)} workgroups")
dur = collections.defaultdict(list)
by_xcd = collections.defaultdict(list)
for i, a, e, hw, info in rows:
    ps = info & 0xff
    dur[ps].append(e - a)
    by_xcd[(hw >> 32) & 0xf].append((a - t0, e - t0, ps, (hw >> 8) & 0xf, (hw >> 13) & 7))
for ps in sorted(dur):
    dd = sorted(dur[ps])
    print(f"pass {ps}: {len(dd)} workgroups, item loop median {dd[len(dd) // 2]} cycles (min {dd[0]}, max {dd[-1]})")
# (s_memtime bases differ between XCDs AND between the shader engines of an XCD: only times of ONE CU are compared with each other)
for x in sorted(by_xcd):
    ws_ = by_xcd[x]
    cus = collections.defaultdict(int); first = {}; last = {}
    for a, e, ps, cu, se in ws_:
        k = (se, cu)
        cus[k] += e - a; first[k] = min(first.get(k, a), a); last[k] = max(last.get(k, e), e)
    spans = sorted(last[k] - first[k] for k in cus)
    cnt = collections.Counter(ps for *_, ps, _c, _s in ws_)
    print(f"XCD {x}: {len(ws_)} workgroups {dict(sorted(cnt.items()))} on {len(cus)} CUs; per CU, first start to last end: min {spans[0]} median {spans[len(spans) // 2]} max {spans[-1]}; "
          f"sum of its item loops: min {min(cus.values())} median {sorted(cus.values())[len(cus) // 2]} max {max(cus.values())}")
if os.environ.get("SHOW_CU"):
    x = sorted(by_xcd)[0]
    key = None
    tl = collections.defaultdict(list)
    for a, e, ps, cu, se in by_xcd[x]:
        tl[(se, cu)].append((a, e, ps))
    key = sorted(tl)[0]
    base = min(a for a, *_ in by_xcd[x])
    print(f"XCD {x} CU {key}: start, end of the item loop (cycles from the XCD's first start), pass, gap to the previous end")
    prev = None
    for a, e, ps in sorted(tl[key]):
        print(f"   {a - base:10d} {e - base:10d}  pass {ps}  loop {e - a:8d}  gap {'' if prev is None else a - prev}")
        prev = e
