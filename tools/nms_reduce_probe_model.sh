set -e
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I r3det-pytorch_amd/csrc -I include -o tools/probes/nms_reduce_probe tools/probes/nms_reduce_probe.hip 2>&1 | grep -i error || true
DUMP_POOL=/tmp/pool.bin python tools/cross_label_edges.py 2>&1 | grep -A2 "image 0"
N=$(python3 -c "import numpy as np; print(int(np.fromfile('/tmp/pool.bin', np.float32, 1)[0]))")
echo N=$N
for g in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do echo "group $g"; tools/probes/nms_reduce_probe $N /tmp/pool.bin $g | tail -8; done
