#!/bin/bash
# Build tools/exp/libs/libfyrox_hip_r04stamp.so: the product library with wall_clock64 stamps (100 MHz) in pose_update_body, read
# by tools/exp/r04_stamps.py.  The product sources are not touched: a copy under /tmp is patched.
# Stamps of the update workgroup's thread 0: 0 entry, 1 top-of-kernel requests issued + program classified, 2 the sampler's
# workgroups have reported (one-launch frames; else = 1), 3 fold done (first node pass), 4 local matrices in LDS (barrier passed),
# 5 hierarchy walk done, 6 matrices copied out, 7 palette stores issued, 8 all stores acknowledged.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf /tmp/var && mkdir -p /tmp/var/fyrox_amd && cp -r "$ROOT/fyrox_amd/csrc" /tmp/var/fyrox_amd/ && cp -r "$ROOT/include" /tmp/var/
cd /tmp/var/fyrox_amd/csrc
python3 - <<'PY'
p = 'anim_kernels.hip'
s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old)
    s = s.replace(old, new, 1)
rep("""    const size_t inst_base = (size_t)inst * rig.n_nodes;

    // Everything that does not depend on the fold""", """    const size_t inst_base = (size_t)inst * rig.n_nodes;
    uint64_t stamp[10];
    for (int q = 0; q < 10; ++q) stamp[q] = 0;
#define STAMP(i) stamp[i] = wall_clock64()
    STAMP(0);

    // Everything that does not depend on the fold""")
rep("""    for (uint32_t node_base = 0; node_base < rig.n_nodes; node_base += bdim) {   // workgroup-uniform trip count""", """    STAMP(1);
    STAMP(2);
    for (uint32_t node_base = 0; node_base < rig.n_nodes; node_base += bdim) {   // workgroup-uniform trip count""")
rep("""                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the records are read behind this, from where the samplers put them
""", """                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the records are read behind this, from where the samplers put them
                STAMP(2);
""")
rep("""            if (cx.dirty && live) {
                trs[0] =""", """            if (node_base == 0) STAMP(3);
            if (cx.dirty && live) {
                trs[0] =""")
rep("""    sync();

    // level-synchronous""", """    sync();
    STAMP(4);

    // level-synchronous""")
rep("""    f4* gout = reinterpret_cast<f4*>(f.global + inst_base * 16);""", """    STAMP(5);
    f4* gout = reinterpret_cast<f4*>(f.global + inst_base * 16);""")
rep("""    for (uint32_t p = 0; p < rig.n_pal; ++p) {
        const PaletteOutDev po = pal_mem ? pal_mem[p] : rig.pal[p];""", """    STAMP(6);
    for (uint32_t p = 0; p < rig.n_pal; ++p) {
        const PaletteOutDev po = pal_mem ? pal_mem[p] : rig.pal[p];""")
# the sampler's workgroup 0, thread 0 (node 0's position.x curve of animation 0): s0 entry, s1 descriptor / time / tick flag here,
# s2 hint here, s3 span record here and the value formed, s4 record store issued, s5 release fence done, s6 barrier passed,
# s7 counter added to (one-launch frames only for s5 - s7)
rep("""__device__ __forceinline__ float sample_curve(""", """__device__ unsigned long long g_sst[16];
#define SSTAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_sst[i] = wall_clock64(); } while (0)
__device__ __forceinline__ float sample_curve(""")
rep("""    if (d.spans && hint >= 1 && hint < d.n_keys) {
        const uint32_t stride = need == 4 ? 16u : 8u;
        const f4* r = reinterpret_cast<const f4*>(d.spans) + (size_t)(hint - 1) * stride;
        f4 locs = r[0];
        if (locs.x < time && time < locs.y) {
            v = interpolate_loaded(locs.x, locs.y, r[1 + 2 * c], r[2 + 2 * c], time);
            sampled = true;""", """    if (d.spans && hint >= 1 && hint < d.n_keys) {
        SSTAMP(2);
        const uint32_t stride = need == 4 ? 16u : 8u;
        const f4* r = reinterpret_cast<const f4*>(d.spans) + (size_t)(hint - 1) * stride;
        f4 locs = r[0];
        if (locs.x < time && time < locs.y) {
            v = interpolate_loaded(locs.x, locs.y, r[1 + 2 * c], r[2 + 2 * c], time);
            if (v == v) SSTAMP(3);
            sampled = true;""")
rep("""    const uint32_t node = (bx * 256u + threadIdx.x) >> 4;
    {""", """    const uint32_t node = (bx * 256u + threadIdx.x) >> 4;
    SSTAMP(0);
    {""")
rep("""        if (!(f.ticked[(size_t)inst * f.n_anims + a] & 1u)) return;      // uniform across the block
""", """        if (!(f.ticked[(size_t)inst * f.n_anims + a] & 1u)) return;      // uniform across the block
        if (time == time && d.present != 0xffffffffu) SSTAMP(1);
""")
rep("""            else *dst = out;
        }
""", """            else *dst = out;
        }
        SSTAMP(4);
""")
rep("""        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(fs.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
""", """        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (blockIdx.x == 0 && threadIdx.x == 0) g_sst[5] = wall_clock64();
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) g_sst[6] = wall_clock64();
        if (threadIdx.x == 0) { const uint32_t was = __hip_atomic_fetch_add(fs.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (blockIdx.x == 0 && was != 0xffffffffu) g_sst[7] = wall_clock64(); }
""")
END = "}\n\n// Two kernel-argument shapes:"
rep(END, """    STAMP(7);
    __builtin_amdgcn_s_waitcnt(0);
    STAMP(8);
    if (threadIdx.x == 0) for (int q = 0; q < 9; ++q) reinterpret_cast<uint64_t*>(f.local + inst_base * 16)[q] = stamp[q];
    if (threadIdx.x == 0) for (int q = 0; q < 8; ++q) reinterpret_cast<uint64_t*>(f.local + inst_base * 16)[9 + q] = __hip_atomic_load(&g_sst[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
""" + END)
open(p, 'w').write(s)
PY
make 2>&1 | grep -E "error" && exit 1
mkdir -p "$ROOT/tools/exp/libs" && cp ../libfyrox_hip.so "$ROOT/tools/exp/libs/libfyrox_hip_r04stamp.so"
echo built
