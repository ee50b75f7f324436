#!/bin/bash
# Build tools/exp/libs/libfyrox_hip_updstamp.so: the product library with eight wall_clock64 stamps in pose_update_body (read by
# tools/exp/upd_stamps.py).  The product sources are not touched: the patch is applied to a copy under /tmp.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf /tmp/var && mkdir -p /tmp/var/fyrox_amd && cp -r "$ROOT/fyrox_amd/csrc" /tmp/var/fyrox_amd/ && cp -r "$ROOT/include" /tmp/var/
cd /tmp/var/fyrox_amd/csrc
python3 - <<'PY'
p = 'anim_kernels.hip'
s = open(p).read()
def rep(old, new, optional=False):
    global s
    if optional and old not in s:
        return
    assert old in s, old
    s = s.replace(old, new, 1)
rep("""    const size_t inst_base = (size_t)inst * rig.n_nodes;
""", """    const size_t inst_base = (size_t)inst * rig.n_nodes;
    uint64_t stamp[10];
    for (int q = 0; q < 10; ++q) stamp[q] = 0;
#define STAMP(i) stamp[i] = wall_clock64()
    STAMP(0);
    const uint64_t cyc0 = clock64();
""")
rep("""    for (uint32_t node_base = 0; node_base < rig.n_nodes; node_base += blockDim.x) {   // workgroup-uniform trip count""", """    STAMP(1);
    for (uint32_t node_base = 0; node_base < rig.n_nodes; node_base += blockDim.x) {   // workgroup-uniform trip count""")
rep("""            if (cx.dirty && live) {
                trs[0] =""", """            STAMP(2);
            if (cx.dirty && live) {
                trs[0] =""")
rep("""    __syncthreads();

    // level-synchronous""", """    __syncthreads();
    STAMP(3);

    // level-synchronous""")
rep("""    for (uint32_t lv = 0; lv < rig.n_levels; ++lv) {
        const uint32_t b2""", """    uint64_t lvc[8];
    for (int q = 0; q < 8; ++q) lvc[q] = 0;
    for (uint32_t lv = 0; lv < rig.n_levels; ++lv) {
#pragma unroll
        for (int q = 0; q < 8; ++q) if (lv == (uint32_t)q) lvc[q] = clock64();
        const uint32_t b2""", optional=True)     # (the walk of tools/exp/r03_walk16.patch: cycles between the starts of its levels)
rep("""    f4* gout = reinterpret_cast<f4*>(f.global + inst_base * 16);""", """    STAMP(4);
    f4* gout = reinterpret_cast<f4*>(f.global + inst_base * 16);""")
if "    const auto palette_column" in s:
    rep("""    const auto palette_column""", """    STAMP(5);
    const auto palette_column""")
else:
    rep("""    for (uint32_t p = 0; p < rig.n_pal; ++p) {
        const PaletteOutDev po = rig.pal[p];""", """    STAMP(5);
    for (uint32_t p = 0; p < rig.n_pal; ++p) {
        const PaletteOutDev po = rig.pal[p];""")
# the end of pose_update_body: the closing brace ahead of the kernel's definition
END = "}\n\ntemplate <bool PROGRAM>\n__global__ __launch_bounds__(256) void pose_update_kernel"
assert s.count(END) == 1
s = s.replace(END, """    STAMP(6);
    __builtin_amdgcn_s_waitcnt(0);
    STAMP(7);
    stamp[8] = clock64() - cyc0;      // shader-clock cycles between stamps 0 and 7
    if (threadIdx.x == 0) for (int q = 0; q < 9; ++q) reinterpret_cast<uint64_t*>(f.local + inst_base * 16)[q] = stamp[q];
    if (threadIdx.x == 0) for (int q = 0; q < 8; ++q) reinterpret_cast<uint64_t*>(f.local + inst_base * 16)[9 + q] = lvc[q];
""" + END, 1)
if "uint64_t lvc[8];" not in s:      # the product's walk: no per-level stamps
    s = s.replace("    STAMP(6);", "    uint64_t lvc[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n    STAMP(6);", 1)
open(p, 'w').write(s)
PY
make 2>&1 | grep -E "error" && exit 1
mkdir -p "$ROOT/tools/exp/libs" && cp ../libfyrox_hip.so "$ROOT/tools/exp/libs/libfyrox_hip_updstamp.so"
echo built
