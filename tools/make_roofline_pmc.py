"""profiles/roofline_kernel_pmc.json from the PMC passes of the roofline kernel (tools/make_profiles.sh step 6):
HBM bytes per launch = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, tagged with the sha256
of the kernel source so that bench.py reports `traffic` only for the code it was measured on.
    python tools/make_roofline_pmc.py profiles/r03_fr_nhwc_pmc.txt [out.json]"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_fr_nhwc_pmc.txt")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "roofline_kernel_pmc.json")
txt = open(src).read()


def counter(name):
    return float(re.search(r"'%s': ([0-9.]+)" % name, txt).group(1))


fetch, write = counter("FETCH_SIZE"), counter("WRITE_SIZE")
hit, miss = counter("TCC_HIT_sum"), counter("TCC_MISS_sum")
sym = re.search(r"(fr_forward_nhwc\w*<[^>]*>)", txt).group(1)
launches = int(re.search(r"n= (\d+)", txt).group(1))
sha = hashlib.sha256(open(os.path.join(ROOT, "r3det-pytorch_amd", "csrc", "r3_fr.hip"), "rb").read()).hexdigest()[:16]
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:  # noqa: BLE001  (the GPU box has no .git: the sha of the source is the tag that matters)
    commit = "gpu-box snapshot"
N, C, H = 4, 256, 128
alg = 16 * N * C * H * H + 20 * N * H * H
hbm = fetch * 1024 * 2 + write * 1024
rec = {
    "kernel_symbol": sym,
    "kernel": "the FeatureRefineModule tail at level 0 on channels_last memory (N=4, C=256, 128x128): (conv_a + bias) + "
              "(conv_b + bias), sampler, residual; 3 reads + 1 write per element",
    "kernel_source_sha16": sha, "commit": commit,
    "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write, "fetch_correction": 2.0,
    "hbm_bytes_per_launch": int(hbm), "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round(hbm / alg, 3), "l2_hit_rate": round(hit / (hit + miss), 3),
    "source": "%s = tools/pmc_groups.sh on tools/fr_nhwc_prof.py (MI355X; separate --pmc passes with --kernel-trace "
              "only; FETCH_SIZE x 2 on gfx950 per MI355X_MICROARCH.md; buffers rotate over 0.8 GB so the launches run "
              "from HBM); %d launches averaged" % (os.path.relpath(src, ROOT), launches),
}
json.dump(rec, open(dst, "w"), indent=1)
print(rec["hbm_bytes_per_launch"], rec["traffic_over_algorithmic"], sym, sha, commit)
