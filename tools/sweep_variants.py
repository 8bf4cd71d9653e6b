#!/usr/bin/env python3
"""Dev tool (GPU box): time every tiling variant of the encoder fast path and check its parity.

For each EEM_V<cin>_<cout>=<idx> setting a subprocess runs (a) the 128x192 golden forward and (b)
eemflow_time_kernels at 1280x720; prints one line per variant.  Usage: python tools/sweep_variants.py
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = {"16_16": [0, 100, 101, 102, 103], "16_32": [0, 100, 101, 102], "32_32": [0, 100, 101, 102],
            "32_64": [0, 100, 101], "64_64": [0, 100, 101, 102]}

CHILD = r'''
import ctypes, json, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from eemflow_amd import EEMFlow, _lib
from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair
g = np.load(os.path.join(%r, "tests/golden/eemflow_fwd_128x192.npz"))
net = EEMFlow("", 5, 5).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(int(g["seed"])).items()})
net = net.cuda(); net.change_imagesize((128, 192))
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(int(g["input_seed"]), 2, 128, 192))
with torch.no_grad():
    flow = net(e1, e2)[1][0]
err = float((flow.cpu() - torch.from_numpy(g["flow"])).abs().max())
net.change_imagesize((720, 1280))
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(3, 1, 720, 1280))
ctx = net._context(e1.device)
out = torch.empty(1, 2, 720, 1280, device="cuda")
stats = (_lib.KernelStat * 64)(); n = ctypes.c_int(0)
L = _lib.lib()
for _ in range(2):
    _lib.check(L.eemflow_time_kernels(ctx, e1.data_ptr(), e2.data_ptr(), 1, 720, 1280, out.data_ptr(), 720, 1280, 30,
                                      stats, 64, ctypes.byref(n), None))
torch.cuda.synchronize()
print(json.dumps({"err": err, "k": {stats[i].name.decode(): round(stats[i].ms * 1e3, 2) for i in range(n.value)}}))
''' % (REPO, REPO)


def run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        return {"err": float("nan"), "k": {}, "fail": r.stderr[-400:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    fams = sys.argv[1:] or list(FAMILIES)
    for fam in fams:
        for idx in FAMILIES[fam]:
            res = run({f"EEM_V{fam}": str(idx)})
            cin, cout = fam.split("_")
            ks = {k: v for k, v in res["k"].items() if k.startswith("enc.") and f"{cin}->{cout}" in k}
            print(f"V{fam}={idx} err={res['err']:.2e} enc_sum={sum(v for k, v in res['k'].items() if k.startswith('enc.')):.1f} {ks} "
                  f"{res.get('fail', '')}", flush=True)


if __name__ == "__main__":
    main()
