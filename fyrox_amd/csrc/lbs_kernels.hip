// lbs_kernels.hip -- gfx950 (MI355X / CDNA4) linear-blend skinning kernels.
//
// What is computed (reference semantics, file:line relative to the Fyrox repo):
//   position : out = sum_{k=0..3} M[idx_k].transform_point(p) * w_k
//              fyrox-impl/src/scene/mesh/mod.rs:501-522 (Mesh::accurate_world_bounding_box)
//   normal, tangent.xyz : out = sum_k (mat3(M[idx_k]) * v) * w_k ; tangent.w copied
//              fyrox-material/src/shader/standard/opengl/standard.shader:187-200
// with nalgebra's operation order: M3x3*v is a column-axpy ((m_i0*x + m_i1*y) + m_i2*z),
// transform_point adds the translation, then divides by the homogeneous n when n != 0.
//
// Hardware mapping (HBM-bound stream: 60 B read + 40 B written per vertex, ~300 VALU):
//   * inputs/outputs are attribute streams in HBM; every lane access is 12- or 16-byte and
//     lanes are contiguous, so each wave instruction covers one dense 768 B / 1 KiB span;
//   * the bone palette of the current instance is staged ONCE per workgroup into LDS as
//     three float4 per bone laid out for packed-f32 math (see stage_palette; 48-byte stride:
//     3 is coprime with the 16 b128 slots, so a random bone gather spreads over all LDS
//     banks) + a separate row-3 array that only the projective (non-affine) path reads;
//   * persistent grid (blocks_per_cu x 256 CUs), each workgroup owns a contiguous range of
//     vertex chunks, re-staging the palette only when the instance changes (crowds);
//   * vertex loads of a chunk are issued BEFORE the palette staging barrier so the HBM
//     latency of the first chunk overlaps the L2->LDS staging;
//   * EXACT=true keeps the reference's unfused operation order (the library is compiled with
//     -ffp-contract=off), making the GPU result bit-identical to the CPU path; EXACT=false
//     uses explicit FMAs (<= 1e-5 relative).  f32 VALU only -- no MFMA: this is not a dense
//     contraction (2.6 FLOP/B).
#include "fyx_internal.h"
#include <hip/hip_ext.h>
#include "lbs_leaves.h"

// A launch that carries its own start / stop events when the caller asked for per-dispatch timing (option lbs.timing:
// the events take the dispatch's begin / end timestamps, what rocprofv3 --kernel-trace reports), a plain launch otherwise.
#define FYX_LAUNCH(t, kernel, grid, block, lds, s, ...)                                                          \
    do {                                                                                                          \
        if ((t).ev_start) hipExtLaunchKernelGGL(kernel, grid, block, lds, s, (t).ev_start, (t).ev_stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__);                                        \
    } while (0)

namespace fyx {


// ---------------------------------------------------------------------------------------
// The skinning kernel.
//
// Work unit = 64 consecutive vertices of one instance (one wave, one vertex per lane; lane
// accesses are 12 B pos/normal, 16 B tangent/weights, 4 B indices, all lane-contiguous, so every
// wave instruction covers one dense, 128-byte-aligned 768 B / 1 KiB / 256 B span).
// Units are dealt out evenly: workgroup b owns the contiguous unit range
// [b*T/G, (b+1)*T/G) -- at most one unit (64 vertices) of imbalance between workgroups, so every
// CU streams the same number of bytes -- and its waves take units round-robin inside that range.
// A range that crosses an instance boundary (few instances of a mesh) is processed per instance
// segment; the palette is (re)staged into LDS once per segment.
//
// Each wave is software-pipelined: the loads of its next unit are issued before the ~300 VALU + 12 LDS
// operations of the current one, so the wave always has a unit in flight in the memory system while it
// computes (all waves start in lock-step at kernel launch, so without this the whole chip alternates
// between a load phase and a compute phase).  Loads and stores are non-temporal: every byte is touched once.
// Measured and left out in rounds 1 - 2 (history: commit e6d81f8, tools/exp/README.md): workgroups of 256 / 1024
// threads, no prefetch / two units ahead / two register sets swapping roles, cacheable accesses, per-wave
// contiguous shares, asymmetric shares and a raised priority for the second-dispatched workgroup of a CU.
// ---------------------------------------------------------------------------------------

constexpr int kSkinBlock = 512;   // threads per workgroup of lbs_skin / lbs_skin_batch / the crowd kernel's vertex tile

template <bool EXACT, int MASK>
__global__ __launch_bounds__(kSkinBlock) void lbs_skin(LbsArgs a, uint32_t units_per_inst, uint32_t total_units) {
    constexpr bool NT = true;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    f32x4* row3 = rows + 3 * a.n_bones;
    uint32_t* wave_flag = reinterpret_cast<uint32_t*>(row3 + a.n_bones);

    constexpr uint32_t WPB = kSkinBlock / 64;
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    if (u_begin >= u_end) return;

    const uint32_t inst_first = u_begin / units_per_inst;
    const uint32_t inst_last = (u_end - 1) / units_per_inst;

    for (uint32_t inst = inst_first; inst <= inst_last; ++inst) {  // workgroup-uniform
        const uint32_t inst_u0 = inst * units_per_inst;
        const uint32_t seg_b = (u_begin > inst_u0 ? u_begin : inst_u0) - inst_u0;
        const uint32_t seg_e = (u_end < inst_u0 + units_per_inst ? u_end : inst_u0 + units_per_inst) - inst_u0;

        // Order of issue matters: vector-memory loads retire in order (vmcnt), so the palette
        // fetch goes FIRST and the first unit's vertex loads right behind it.  The LDS commit
        // below then waits for the (L2-resident, fast) palette only, while the vertex loads --
        // a cold HBM access at kernel start -- stay in flight across the staging barrier.
        const PaletteRegs pr = palette_fetch(a.palette + (size_t)inst * a.n_bones * 16, a.n_bones, tid);
        // this wave's vertices of the segment: whole units dealt round-robin to the waves (unit k of the segment -> wave k % WPB)
        const uint32_t ve = seg_e * 64 < a.n_verts ? seg_e * 64 : a.n_verts;
        constexpr uint32_t vstep = WPB * 64;
        uint32_t base = (seg_b + wave) * 64;
        uint32_t v = base + lane;
        VertexIn<MASK> cur = load_vertex<NT, MASK>(a, v < ve ? v : 0);

        if (inst != inst_first) __syncthreads();  // every wave is done with the previous palette
        const bool pj = palette_commit(pr, a.n_bones, rows, row3, tid);
        // block-wide OR with the single barrier the staging needs anyway: one flag per wave
        const bool wave_pj = __any(pj) != 0;
        if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;
        __syncthreads();
        bool projective = false;
#pragma unroll
        for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;
        // Opaque use point: nothing that consumes the first unit's vertex data may be scheduled
        // above the staging barrier (it would drag the wait for those loads up with it).
        pin_vertex(cur);

        while (base < ve) {  // wave-uniform
            const uint32_t bn = base + vstep;
            const uint32_t vn = bn + lane;
            VertexIn<MASK> nxt;
            if (bn < ve) nxt = load_vertex<NT, MASK>(a, vn < ve ? vn : 0);
            const Skinned o = skin_vertex<EXACT, MASK>(rows, row3, projective, cur.id, cur.w, cur.p.x,
                                                       cur.p.y, cur.p.z, cur.n.x, cur.n.y, cur.n.z,
                                                       cur.t.x, cur.t.y, cur.t.z);
            if (v < ve) {
                const size_t ov = (size_t)inst * a.n_verts + v;
                if constexpr (MASK & 1) st3<NT>(a.out_pos + ov * 3, o.px, o.py, o.pz);
                if constexpr (MASK & 2) st3<NT>(a.out_nrm + ov * 3, o.nx, o.ny, o.nz);
                if constexpr (MASK & 4)
                    stg<NT>(reinterpret_cast<f32x4*>(a.out_tan) + ov, f32x4{o.tx, o.ty, o.tz, cur.t.w});
            }
            cur = nxt;
            base = bn;
            v = vn;
        }
    }
}


// ---------------------------------------------------------------------------------------
// lbs_skin_dyn: the single-instance kernel for large meshes, built around how a LONE launch spends its ~18 us
// (timeline study of round 2: profiles/r02_lone_launch/, profiles/r01_timeline.json for lbs_skin).
//
//   * Four workgroups of 256 threads per CU -- all 16 waves a CU holds at this register budget, resident anyway -- each
//     owning one contiguous range of 64-vertex units.  Inside the workgroup the waves DRAW their units from an LDS ticket
//     counter (one ds_add_rtn per unit) instead of taking every WPB-th one: a wave that is served faster takes more, so
//     the workgroup ends when its work does, not when the wave that happened to get 4 units instead of 3 does (measured
//     spread inside a workgroup of lbs_skin: 1.9 us of a 16 us launch).  (Drawing chunks from device-scope counters shared
//     by all CUs was measured too: a returning global atomic queues behind the CU's own streaming loads, several us each
//     -- 27 us per launch at best.  Nothing here leaves the CU.)
//   * The palette goes first and wide: every thread fetches 16-byte columns of the palette (one dense 1 KB request per
//     wave, the wave's FIRST vector-memory instruction) and the first unit's vertex loads right behind it; the second
//     unit is requested as soon as the staging barrier is passed.  (Requesting both units ahead of the barrier was
//     measured slower, 20.0 vs 19.5 us: the CU's memory queue is served in order, so the palette columns of the later
//     waves then wait behind twice as many vertex requests of the earlier ones.)
//   * The loop keeps two units in flight per wave: a register set is refilled as soon as its math is done, while the
//     other set's loads have had a whole unit's time to land.
//   * The streams are buffer resources: a lane past the end of the mesh loads zeros and its stores are dropped by the
//     hardware, so every unit is five loads and three stores of straight-line vector-memory code, and the cache policy
//     rides in the instruction: loads `nt` (read once), stores `sc1` (written through, the line is not kept dirty in the
//     XCD's L2).  Measured over all 25 load / store policy pairs (profiles/r02_policy_sweep_*.json): sc1 stores are
//     0.5 us faster than nt stores on a lone launch and 1.0 us (6 %) faster when launches overlap on two streams.
// Which wave skins a unit changes nothing in the arithmetic: results are bit-identical to lbs_skin's.
// The experiment switches this kernel carried in round 2 (start-up pauses, unit pools with scalar-memory tickets, partner
// stealing, unit-tiled inputs, ablations, 512 / 1024-thread workgroups, five- / six-wave register budgets, per-wave
// timeline probe) are in the history (commit e6d81f8) with their results in DESIGN.md 5.
// ---------------------------------------------------------------------------------------
constexpr int kDynBlock = 256;
constexpr int kDynLoadAux = 2, kDynStoreAux = 16;     // nt loads, sc1 stores

template <bool EXACT, int MASK>
__global__ __launch_bounds__(kDynBlock) void lbs_skin_dyn(LbsArgs a, uint32_t total_units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    f32x4* row3 = rows + 3 * a.n_bones;
    uint32_t* wave_flag = reinterpret_cast<uint32_t*>(row3 + a.n_bones);   // 16 words
    uint32_t* ticket = wave_flag + 16;

    constexpr uint32_t WPB = kDynBlock / 64;
    constexpr int PIECES = 1024 / kDynBlock;     // 16-byte palette columns per thread (n_bones <= 256)
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t n_units = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x) - u_begin;

    // Palette columns first: column c of bone b is piece 4 b + c, and a wave fetches 64 consecutive pieces (16 bones, one dense
    // 1 KB request).  WHICH lane takes which piece of those 64 is chosen for the LDS commit below: the 16 lanes one ds_write_b64
    // serves together take the four columns of bones {0, 2, 4, 6} (or {1, 3, 5, 7}) of an octet, whose 8-dword (x, y) windows at
    // the 48-byte row stride start at dwords 0 / 24 / 48 / 72 = banks 0 / 24 / 16 / 8: the four windows tile all 32 banks.  (With
    // lanes in piece order the group was bones {0, 1, 2, 3}: windows at banks 0 / 12 / 24 / 4, the fourth on top of the first --
    // SQ_LDS_BANK_CONFLICT 65 536 cycles per launch in round 2.)
    const uint32_t n_pieces = a.n_bones * 4;
    const uint32_t grp16 = lane >> 4, in16 = lane & 15;
    const uint32_t my_piece = ((uint32_t)tid & ~63u) + 4 * (8 * (grp16 >> 1) + 2 * (in16 >> 2) + (grp16 & 1)) + (in16 & 3);
    f32x4 col[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const uint32_t piece = my_piece + (uint32_t)i * kDynBlock;
        col[i] = reinterpret_cast<const f32x4*>(a.palette)[piece < n_pieces ? piece : n_pieces - 1];
    }
    // the wave's first two units (tickets wave and WPB + wave; the launcher guarantees n_units >= 2 WPB)
    const VtxBuffers vb = make_vtx_buffers(a);
    auto vertex_of = [&](uint32_t t) -> uint32_t { return (u_begin + t) * 64 + lane; };
    uint32_t vA = vertex_of(wave), vB = vertex_of(WPB + wave);
    VertexIn<MASK> A = load_vertex_buf<MASK, kDynLoadAux>(vb, vA);
    VertexIn<MASK> B;   // requested behind the staging barrier (see above)

    if (tid == 0) *ticket = 2 * WPB;
    bool pj = false;
    // Packed-math layout (see stage_palette): A = (m00, m10, m01, m11)  B = (m02, m12, t0, t1)  C = (m20, m21, m22, t2)
    // row3 = (m30, m31, m32, m33); a thread holds column c = (m0c, m1c, m2c, m3c) of bone b: (x, y) is one 8-byte store, z and
    // w one 4-byte store each (the 32 lanes of a 4-byte store cover a whole octet of bones: dwords 12 b + 8 + c, all banks).
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const uint32_t piece = my_piece + (uint32_t)i * kDynBlock;
        if (piece < n_pieces) {
            const uint32_t b = piece >> 2, c = piece & 3;
            float* r = reinterpret_cast<float*>(rows + b * 3);
            *reinterpret_cast<f32x2*>(r + 2 * c) = f32x2{col[i].x, col[i].y};
            r[8 + c] = col[i].z;
            reinterpret_cast<float*>(row3 + b)[c] = col[i].w;
            pj |= col[i].w != (c == 3 ? 1.0f : 0.0f);
        }
    }
    const bool wave_pj = __any(pj) != 0;
    if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;
    __syncthreads();
    bool projective = false;
#pragma unroll
    for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;
    pin_vertex(A);
    B = load_vertex_buf<MASK, kDynLoadAux>(vb, vB);

    auto process = [&](VertexIn<MASK>& c_, uint32_t v_c) {
        pin_vertex(c_);   // the math's first touch of the loaded registers is here
        const Skinned o = skin_vertex<EXACT, MASK>(rows, row3, projective, c_.id, c_.w, c_.p.x, c_.p.y, c_.p.z,
                                                   c_.n.x, c_.n.y, c_.n.z, c_.t.x, c_.t.y, c_.t.z);
        store_vertex_buf<MASK, kDynStoreAux>(vb, v_c, o, c_.t.w);
    };
    // draw the next unit into a register set whose math is done; false when the workgroup's range is used up
    auto refill = [&](VertexIn<MASK>& n_, uint32_t& v_n) -> bool {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= n_units) return false;
        v_n = vertex_of(t);
        n_ = load_vertex_buf<MASK, kDynLoadAux>(vb, v_n);
        return true;
    };
    for (;;) {   // wave-uniform
        process(A, vA);
        if (!refill(A, vA)) { process(B, vB); break; }
        process(B, vB);
        if (!refill(B, vB)) { process(A, vA); break; }
    }
}

// ---------------------------------------------------------------------------------------
// Crowd kernel: many instances of ONE mesh (1000 x 10 k vertices / 64 bones).
//
// lbs_skin above re-reads the shared mesh for every instance (60 B/vertex from L2 on top of the
// 40 B written), which makes a crowd L2/TA-bound.  Here a workgroup owns one tile of BLOCK
// vertices (one per thread) and a run of `ipb` consecutive instances: the vertex attributes are
// loaded ONCE into registers and stay there, and only the palette changes per instance.  The
// palettes are double-buffered in LDS -- the global fetch of palette i+1 is issued before the
// barrier of instance i and committed to the other buffer after this wave's math -- so there is
// exactly one barrier per instance and the only per-instance memory traffic is the palette
// (n_bones * 64 B per BLOCK vertices) and the 40 B/vertex written.
// blockIdx -> (tile, chunk) with tile fastest: neighbouring workgroups (which round-robin over
// the XCDs) read the same palettes, so each palette is fetched from HBM about once per XCD.
//
// What bounds it (C3, 1000 x 10 k / 64 bones; ISA count of the affine path: 69 v_pk_mul + 55 v_pk_add + 16 single mul / add
// + ~20 moves and address updates per vertex-instance; a packed instruction issues in ~1.9 ns per SIMD, a VOP2 one in ~1.1:
// the packed form has the single form's rate per RESULT, it only halves the instruction count): 44 - 46 us of VALU issue
// beside 62 - 65 us of stores.  In the fused mode (~45 VALU) the launch IS the stores: 60 - 67 us, 0.75 - 0.84 of peak.  In
// the exact mode the kernel is bound by its COMPUTE PATH (round 3, profiles/r03_crowd_study/): without its stores it still
// takes 73 - 84 us of 79 - 87; of those the arithmetic and LDS gathers are 49, the barrier 3, and the per-instance palette fetch +
// commit on the one bone-owning wave 21 (waves 0 - 3 finish an instance's arithmetic in 1 360 cycles, waves 4 - 7 -- who lose
// the VALU arbitration by age -- in 2 300, and wave 0 then spends another 1 120 on its palette).  Measured against that and NOT
// kept, all bit-identical (tools/exp/r03_crowd_forms.patch): line-aligned stores through LDS strips, stores after the commit, a
// loading wave that never stores, a persistent grid (also with staggered starts), a ring of eight palettes with one barrier per
// eight instances, palettes copied by LDS-DMA from a pre-packed buffer with a counted vmcnt, a raised priority for waves 4 - 7;
// round 2: two / four instances per barrier, palette columns staged by every thread, per-instance buffer-resource outputs, sc1
// stores.  Compute and the 400 MB of stores each take 60 - 75 us alone and overlap only partly whatever the structure.
// Round 4 (tools/exp/r04_crowd_forms.patch, profiles/r04_crowd_forms.jsonl; one process, forms interleaved, bit-identical): the
// palette path amortised over twice the vertices is no faster either -- two vertices per thread 82.1 us (influences one by one,
// 100 VGPRs) / 86.3 (all rows live, 170 VGPRs, one workgroup per CU), a 1024-thread workgroup 82.3 / 87.9, against 79.4 - 80.1
// for this kernel.  Exact ~0.62 - 0.64 of the HBM roofline is this structure's limit; the fused mode (<= 1e-5, inside north_star's
// tolerance) is the crowd's fast mode.
// ---------------------------------------------------------------------------------------
// LEAN (option lbs.crowd_lean): the arithmetic walks the influences one by one (SEQ above: ~74 VGPRs) and the launch asks
// for enough LDS that only two workgroups share a CU -- four waves per SIMD holding ~320 of its 512 VGPRs, which leaves
// room for the waves of OTHER kernels: what lets the next frame's pose kernels run under this frame's skinning
// (anim.overlap) instead of trickling in as the skinning drains.
template <int BLOCK, bool EXACT, int MASK, bool LEAN = false>
__global__ __launch_bounds__(BLOCK) void lbs_skin_crowd(LbsArgs a, uint32_t tiles, uint32_t ipb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t buf_f4 = 4 * a.n_bones;  // rows (3 per bone) + row3 (1 per bone)
    f32x4* const base = reinterpret_cast<f32x4*>(smem);
    uint32_t* const flags = reinterpret_cast<uint32_t*>(base + 2 * buf_f4);  // [2][4]

    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in blockIdx order, and all `tiles`
    // workgroups of an instance run read the same palettes: ranked by (XCD, order within the XCD) instead of by blockIdx,
    // consecutive ranks -- hence the workgroups of one instance run -- sit on ONE XCD, and a palette comes out of HBM
    // once instead of once per XCD (PMC: 33 MB fetched per C3 launch for 4.7 MB of mesh and palettes).  Any permutation
    // of the workgroups computes the same thing; only the placement is a guess about the hardware.
    const uint32_t G = gridDim.x, x = blockIdx.x & 7u;
    uint32_t rank = blockIdx.x >> 3;
    for (uint32_t q = 0; q < x; ++q) rank += (G - q + 7u) >> 3;     // workgroups on the XCDs before this one
    const uint32_t tile = rank % tiles, chunk = rank / tiles;
    const uint32_t i0 = chunk * ipb;
    const uint32_t i1 = (i0 + ipb < a.n_instances) ? i0 + ipb : a.n_instances;
    if (i0 >= i1) return;
    const uint32_t v = tile * BLOCK + tid;
    const bool live = v < a.n_verts;

    // Only the waves whose threads own a bone (n_bones <= 256: waves 0..3) take part in the palette traffic; each
    // of those four waves owns one flag word per buffer (a wave without bones writes 0 there).
    const bool owner = wave * 64 < a.n_bones;   // wave-uniform
    PaletteRegs pr;
    if (owner) pr = palette_fetch(a.palette + (size_t)i0 * a.n_bones * 16, a.n_bones, tid);
    // the mesh is shared by every workgroup of the launch: ordinary (cacheable) loads
    const VertexIn<MASK> vin = load_vertex<false, MASK>(a, live ? v : 0);
    if (wave < 4) {
        bool wave_pj = false;
        if (owner) wave_pj = __any(palette_commit(pr, a.n_bones, base, base + 3 * a.n_bones, tid)) != 0;
        if (lane == 0) flags[wave] = wave_pj ? 1u : 0u;
    }
    uint32_t cur = 0;
    for (uint32_t inst = i0; inst < i1; ++inst) {  // workgroup-uniform
        const bool more = inst + 1 < i1;
        if (more && owner) pr = palette_fetch(a.palette + (size_t)(inst + 1) * a.n_bones * 16, a.n_bones, tid);
        __syncthreads();  // buffer `cur` is complete; nobody reads buffer `cur ^ 1` any more
        const f32x4* rows = base + cur * buf_f4;
        const f32x4* row3 = rows + 3 * a.n_bones;
        const u32x4 fl = *reinterpret_cast<const u32x4*>(flags + cur * 4);
        const bool projective = (fl.x | fl.y | fl.z | fl.w) != 0;
        // the crowd kernel is VALU-bound, so its fused mode blends the matrices first (see skin_vertex_blended)
        const Skinned o = skin_vertex<EXACT, MASK, true, LEAN>(rows, row3, projective, vin.id, vin.w, vin.p.x, vin.p.y,
                                                               vin.p.z, vin.n.x, vin.n.y, vin.n.z, vin.t.x, vin.t.y, vin.t.z);
        if (live) {
            const size_t ov = (size_t)inst * a.n_verts + v;
            if constexpr (MASK & 1) st3<true>(a.out_pos + ov * 3, o.px, o.py, o.pz);
            if constexpr (MASK & 2) st3<true>(a.out_nrm + ov * 3, o.nx, o.ny, o.nz);
            if constexpr (MASK & 4)
                stg<true>(reinterpret_cast<f32x4*>(a.out_tan) + ov, f32x4{o.tx, o.ty, o.tz, vin.t.w});
        }
        if (more && wave < 4) {
            f32x4* nrows = base + (cur ^ 1) * buf_f4;
            bool wave_pj = false;
            if (owner) wave_pj = __any(palette_commit(pr, a.n_bones, nrows, nrows + 3 * a.n_bones, tid)) != 0;
            if (lane == 0) flags[(cur ^ 1) * 4 + wave] = wave_pj ? 1u : 0u;
        }
        cur ^= 1;
    }
}

template <int BLOCK, bool EXACT, int MASK>
static hipError_t launch_crowd_one(const LbsArgs& a, const LbsTuning& t, hipStream_t s) {
    const uint32_t tiles = (a.n_verts + BLOCK - 1) / BLOCK;
    uint32_t ipb = (uint32_t)(t.crowd_ipb > 0 ? t.crowd_ipb : 0);
    if (ipb == 0) {
        // long enough runs to amortise the vertex loads, enough workgroups to fill the chip.  C3: runs of 8 are 3 - 4 % faster than
        // 16 in the fused mode (round 3, three boxes) and 1 - 5 % in the exact mode (round 4: 79.4 against 80.1 us alone, 72.0
        // against 75.9 - 77.7 us inside the frame loop); 2 and 32 lose ~10 %
        const uint64_t pairs = (uint64_t)tiles * a.n_instances;
        uint64_t want = pairs / ((uint64_t)kCUs * 4);
        if (want < 1) want = 1;
        if (want > 8u) want = 8u;
        ipb = (uint32_t)want;
    }
    if (ipb > a.n_instances) ipb = a.n_instances;
    const uint32_t chunks = (a.n_instances + ipb - 1) / ipb;
    const uint64_t grid = (uint64_t)tiles * chunks;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = (size_t)a.n_bones * 64 * 2 + 2 * (BLOCK / 64) * sizeof(uint32_t);
    if constexpr (BLOCK == 512 && EXACT) {
        if (t.crowd_lean) {   // two workgroups per CU: more than a third of the CU's 160 KB of LDS each
            const size_t lean_lds = std::max<size_t>(lds, 56 * 1024);
            FYX_LAUNCH(t, (lbs_skin_crowd<BLOCK, EXACT, MASK, true>), dim3((uint32_t)grid), dim3(BLOCK), (uint32_t)lean_lds, s, a, tiles, ipb);
            return hipGetLastError();
        }
    }
    FYX_LAUNCH(t, (lbs_skin_crowd<BLOCK, EXACT, MASK>), dim3((uint32_t)grid), dim3(BLOCK), (uint32_t)lds, s, a, tiles, ipb);
    return hipGetLastError();
}

template <int BLOCK, bool EXACT>
static hipError_t launch_crowd_mask(const LbsArgs& a, const LbsTuning& t, hipStream_t s) {
    const int mask = (a.out_pos ? 1 : 0) | ((a.out_nrm && a.nrm) ? 2 : 0) | ((a.out_tan && a.tan) ? 4 : 0);
    switch (mask) {
        case 1: return launch_crowd_one<BLOCK, EXACT, 1>(a, t, s);
        case 2: return launch_crowd_one<BLOCK, EXACT, 2>(a, t, s);
        case 3: return launch_crowd_one<BLOCK, EXACT, 3>(a, t, s);
        case 4: return launch_crowd_one<BLOCK, EXACT, 4>(a, t, s);
        case 5: return launch_crowd_one<BLOCK, EXACT, 5>(a, t, s);
        case 6: return launch_crowd_one<BLOCK, EXACT, 6>(a, t, s);
        case 7: return launch_crowd_one<BLOCK, EXACT, 7>(a, t, s);
        default: return hipSuccess;
    }
}

static hipError_t launch_crowd(const LbsArgs& a, const LbsTuning& t, hipStream_t s) {
    // one tile size: 512 vertices (256 measured equal within 3 % in rounds 1 - 2 and left out)
    return t.exact ? launch_crowd_mask<kSkinBlock, true>(a, t, s) : launch_crowd_mask<kSkinBlock, false>(a, t, s);
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
template <bool EXACT, int MASK>
static hipError_t launch_one(const LbsArgs& a, const LbsTuning& t, hipStream_t s) {
    const uint32_t upi = (a.n_verts + 63) / 64;
    const uint64_t total64 = (uint64_t)upi * a.n_instances;
    if (total64 == 0) return hipSuccess;
    if (total64 > 0xffffffffull) return hipErrorInvalidValue;
    const uint32_t total = (uint32_t)total64;
    uint32_t grid = (uint32_t)kCUs * (uint32_t)(t.blocks_per_cu > 0 ? t.blocks_per_cu : 1);
    const uint32_t max_useful = (total + (kSkinBlock / 64) - 1) / (kSkinBlock / 64);
    if (grid > max_useful) grid = max_useful;
    const size_t lds = (size_t)a.n_bones * 64 + 64;  // rows + row3 + one flag per wave
    FYX_LAUNCH(t, (lbs_skin<EXACT, MASK>), dim3(grid), dim3(kSkinBlock), (uint32_t)lds, s, a, upi, total);
    return hipGetLastError();
}

// lbs_skin_dyn launch: the grid is what is resident (16 waves per CU at the kernel's register budget).  Returns
// hipErrorNotReady when the launch does not qualify (the caller takes lbs_skin).
template <bool EXACT, int MASK>
static hipError_t launch_dyn_one(const LbsArgs& a, const LbsTuning& t, hipStream_t s) {
    constexpr uint32_t WPB = kDynBlock / 64;
    const uint32_t total = (a.n_verts + 63) / 64;
    const uint32_t grid = (uint32_t)kCUs * (1024u / kDynBlock);
    if (total / grid < 2 * WPB) return hipErrorNotReady;   // every wave starts with two units of its own
    const size_t lds = (size_t)a.n_bones * 64 + 64 + 16;
    FYX_LAUNCH(t, (lbs_skin_dyn<EXACT, MASK>), dim3(grid), dim3(kDynBlock), (uint32_t)lds, s, a, total);
    return hipGetLastError();
}

template <bool EXACT, bool DYN>
static hipError_t launch_mask(const LbsArgs& a, const LbsTuning& t, hipStream_t s) {
    const int mask = (a.out_pos ? 1 : 0) | ((a.out_nrm && a.nrm) ? 2 : 0) | ((a.out_tan && a.tan) ? 4 : 0);
#define FYX_MASK_CASE(M) case M: if constexpr (DYN) return launch_dyn_one<EXACT, M>(a, t, s); else return launch_one<EXACT, M>(a, t, s);
    switch (mask) {
        FYX_MASK_CASE(1) FYX_MASK_CASE(2) FYX_MASK_CASE(3) FYX_MASK_CASE(4) FYX_MASK_CASE(5) FYX_MASK_CASE(6) FYX_MASK_CASE(7)
        default: return hipSuccess;  // nothing requested
    }
#undef FYX_MASK_CASE
}

hipError_t launch_lbs(const LbsArgs& a, const LbsTuning& t, hipStream_t stream) {
    if (a.n_verts == 0 || a.n_instances == 0) return hipSuccess;
    if (t.dyn && a.n_instances == 1 && a.n_bones != 0 && a.n_bones <= 256 && a.n_verts <= 0x0fffffffu) {
        const hipError_t e = t.exact ? launch_mask<true, true>(a, t, stream) : launch_mask<false, true>(a, t, stream);
        if (e != hipErrorNotReady) return e;
    }
    // crowds (one mesh, many palettes) keep the vertices in registers and loop over instances
    if (t.crowd > 0 || (t.crowd < 0 && a.n_instances >= 4)) return launch_crowd(a, t, stream);
    return t.exact ? launch_mask<true, false>(a, t, stream) : launch_mask<false, false>(a, t, stream);
}

// ---------------------------------------------------------------------------------------
// Batched skinning: many (mesh, palette) pairs in one launch (fyx_lbs_skin_batch).
//
// A scene of a few hundred distinct characters is a few hundred launches of a few microseconds of work each; the
// GPU idles between them (measured: 256 meshes of 5 k vertices, 1.0 ms of launches for 13 us of HBM traffic).  Here
// the jobs' instances ("segments") are laid end to end in one numbering of 64-vertex units and the launch splits
// THAT evenly over its workgroups, as lbs_skin does for the instances of one mesh: a workgroup walks its unit range
// segment by segment, restaging the palette at each boundary.  The segment's pointers come from a table in HBM read
// through the constant address space (uniform address => scalar loads into SGPRs, as kernel arguments would be).
// Same per-vertex code as lbs_skin (load_vertex / skin_vertex / the stores): bit-identical results.
// ---------------------------------------------------------------------------------------
#define FYX_CONSTANT __attribute__((address_space(4)))

template <bool EXACT, int MASK>
__global__ __launch_bounds__(512) void lbs_skin_batch(const LbsSegDev* __restrict__ segs_g, uint32_t n_segs,
                                                     const uint32_t* __restrict__ block_seg, uint32_t total_units) {
    constexpr uint32_t WPB = 8;
    constexpr bool NT = true;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FYX_CONSTANT LbsSegDev* segs = (const FYX_CONSTANT LbsSegDev*)segs_g;
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    if (u_begin >= u_end) return;
    bool first = true;
    for (uint32_t seg = block_seg[blockIdx.x]; seg < n_segs; ++seg) {   // workgroup-uniform
        const uint32_t s_u0 = segs[seg].unit0;
        if (s_u0 >= u_end) break;
        LbsArgs a;
        a.pos = segs[seg].pos; a.nrm = segs[seg].nrm; a.tan = segs[seg].tan; a.wgt = segs[seg].wgt; a.idx = segs[seg].idx;
        a.palette = segs[seg].palette;
        a.out_pos = segs[seg].out_pos; a.out_nrm = segs[seg].out_nrm; a.out_tan = segs[seg].out_tan;
        a.n_verts = segs[seg].n_verts; a.n_bones = segs[seg].n_bones; a.n_instances = 1;
        const uint32_t upi = (a.n_verts + 63) / 64;
        if (s_u0 + upi <= u_begin) continue;
        const uint32_t seg_b = (u_begin > s_u0 ? u_begin : s_u0) - s_u0;
        const uint32_t seg_e = (u_end < s_u0 + upi ? u_end : s_u0 + upi) - s_u0;

        f32x4* rows = reinterpret_cast<f32x4*>(smem);
        f32x4* row3 = rows + 3 * a.n_bones;
        uint32_t* wave_flag = reinterpret_cast<uint32_t*>(row3 + a.n_bones);
        // palette fetch first, the first unit's vertex loads right behind it (see lbs_skin)
        const PaletteRegs pr = palette_fetch(a.palette, a.n_bones, tid);
        const uint32_t vb = (seg_b + wave) * 64;
        const uint32_t ve = seg_e * 64 < a.n_verts ? seg_e * 64 : a.n_verts;
        constexpr uint32_t vstep = WPB * 64;
        uint32_t base = vb;
        uint32_t v = base + lane;
        VertexIn<MASK> cur = load_vertex<NT, MASK>(a, v < ve ? v : 0);

        if (!first) __syncthreads();  // every wave is done with the previous palette
        first = false;
        const bool pj = palette_commit(pr, a.n_bones, rows, row3, tid);
        const bool wave_pj = __any(pj) != 0;
        if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;
        __syncthreads();
        bool projective = false;
#pragma unroll
        for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;
        pin_vertex(cur);

        while (base < ve) {  // wave-uniform
            const uint32_t bn = base + vstep;
            const uint32_t vn = bn + lane;
            VertexIn<MASK> nxt;
            if (bn < ve) nxt = load_vertex<NT, MASK>(a, vn < ve ? vn : 0);
            const Skinned o = skin_vertex<EXACT, MASK>(rows, row3, projective, cur.id, cur.w, cur.p.x,
                                                       cur.p.y, cur.p.z, cur.n.x, cur.n.y, cur.n.z,
                                                       cur.t.x, cur.t.y, cur.t.z);
            if (v < ve) {
                if constexpr (MASK & 1) st3<NT>(a.out_pos + (size_t)v * 3, o.px, o.py, o.pz);
                if constexpr (MASK & 2) st3<NT>(a.out_nrm + (size_t)v * 3, o.nx, o.ny, o.nz);
                if constexpr (MASK & 4)
                    stg<NT>(reinterpret_cast<f32x4*>(a.out_tan) + v, f32x4{o.tx, o.ty, o.tz, cur.t.w});
            }
            cur = nxt;
            base = bn;
            v = vn;
        }
    }
}

// lbs_skin_batch_dyn (option lbs.dyn, the default): the batch in lbs_skin_dyn's form.  lbs_skin_batch above is lbs_skin's structure -- 512-thread
// workgroups at 104 VGPRs, so only two of a CU's four are resident and the launch runs in two rounds; eight waves that take every eighth
// unit of a workgroup's ~20 (two or three each: the workgroup ends with the waves that got three); one unit requested ahead.  Here: four
// 256-thread workgroups per CU, all resident; the four waves of a workgroup DRAW the units of the workgroup's range from an LDS ticket, two
// units in flight per wave; palette columns fetched by every thread; the streams of a segment as buffer resources (the ragged last unit of a
// mesh needs no bounds test), loads nt, stores sc1.  A workgroup whose range crosses into another mesh does all of that again per segment.
// Same per-vertex code: bit-identical results.  256 meshes x 5 000 vertices / 64 bones (128 MB): see DESIGN 5.
constexpr int kBatchDynBlock = 256;

template <bool EXACT, int MASK>
__global__ __launch_bounds__(kBatchDynBlock) void lbs_skin_batch_dyn(const LbsSegDev* __restrict__ segs_g, uint32_t n_segs,
                                                                     const uint32_t* __restrict__ block_seg, uint32_t total_units) {
    constexpr uint32_t WPB = kBatchDynBlock / 64;
    constexpr int PIECES = 1024 / kBatchDynBlock;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* const wave_flag = reinterpret_cast<uint32_t*>(smem);          // 4 words; word 8: the ticket; the palette from byte 64
    uint32_t* const ticket = wave_flag + 8;
    f32x4* const rows = reinterpret_cast<f32x4*>(smem + 64);
    const FYX_CONSTANT LbsSegDev* segs = (const FYX_CONSTANT LbsSegDev*)segs_g;
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    if (u_begin >= u_end) return;
    const uint32_t grp16 = lane >> 4, in16 = lane & 15;
    const uint32_t my_piece = ((uint32_t)tid & ~63u) + 4 * (8 * (grp16 >> 1) + 2 * (in16 >> 2) + (grp16 & 1)) + (in16 & 3);   // (see lbs_skin_dyn)
    bool first = true;
    for (uint32_t seg = block_seg[blockIdx.x]; seg < n_segs; ++seg) {   // workgroup-uniform
        const uint32_t s_u0 = segs[seg].unit0;
        if (s_u0 >= u_end) break;
        LbsArgs a;
        a.pos = segs[seg].pos; a.nrm = segs[seg].nrm; a.tan = segs[seg].tan; a.wgt = segs[seg].wgt; a.idx = segs[seg].idx;
        a.palette = segs[seg].palette;
        a.out_pos = segs[seg].out_pos; a.out_nrm = segs[seg].out_nrm; a.out_tan = segs[seg].out_tan;
        a.n_verts = segs[seg].n_verts; a.n_bones = segs[seg].n_bones; a.n_instances = 1;
        const uint32_t upi = (a.n_verts + 63) / 64;
        if (s_u0 + upi <= u_begin) continue;
        const uint32_t seg_b = (u_begin > s_u0 ? u_begin : s_u0) - s_u0;
        const uint32_t n_units = (u_end < s_u0 + upi ? u_end : s_u0 + upi) - s_u0 - seg_b;
        f32x4* const row3 = rows + 3 * a.n_bones;

        const uint32_t n_pieces = a.n_bones * 4;
        f32x4 col[PIECES];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const uint32_t piece = my_piece + (uint32_t)i * kBatchDynBlock;
            col[i] = n_pieces ? reinterpret_cast<const f32x4*>(a.palette)[piece < n_pieces ? piece : n_pieces - 1] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const VtxBuffers vb = make_vtx_buffers(a);
        auto vertex_of = [&](uint32_t t) -> uint32_t { return (seg_b + t) * 64 + lane; };
        bool hasA = wave < n_units, hasB = WPB + wave < n_units;      // wave-uniform
        uint32_t vA = vertex_of(wave), vB = vertex_of(WPB + wave);
        VertexIn<MASK> A, B;
        if (hasA) A = load_vertex_buf<MASK, kDynLoadAux>(vb, vA);

        if (!first) __syncthreads();  // every wave is done with the previous segment's palette and ticket
        first = false;
        if (tid == 0) *ticket = 2 * WPB;
        bool pj = false;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const uint32_t piece = my_piece + (uint32_t)i * kBatchDynBlock;
            if (piece < n_pieces) {
                const uint32_t b = piece >> 2, c = piece & 3;
                float* r = reinterpret_cast<float*>(rows + b * 3);
                *reinterpret_cast<f32x2*>(r + 2 * c) = f32x2{col[i].x, col[i].y};
                r[8 + c] = col[i].z;
                reinterpret_cast<float*>(row3 + b)[c] = col[i].w;
                pj |= col[i].w != (c == 3 ? 1.0f : 0.0f);
            }
        }
        const bool wave_pj = __any(pj) != 0;
        if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;
        __syncthreads();
        bool projective = false;
#pragma unroll
        for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;
        if (hasA) pin_vertex(A);
        if (hasB) B = load_vertex_buf<MASK, kDynLoadAux>(vb, vB);

        auto process = [&](VertexIn<MASK>& c_, uint32_t v_c) {
            pin_vertex(c_);
            const Skinned o = skin_vertex<EXACT, MASK>(rows, row3, projective, c_.id, c_.w, c_.p.x, c_.p.y, c_.p.z,
                                                       c_.n.x, c_.n.y, c_.n.z, c_.t.x, c_.t.y, c_.t.z);
            store_vertex_buf<MASK, kDynStoreAux>(vb, v_c, o, c_.t.w);
        };
        auto refill = [&](VertexIn<MASK>& n_, uint32_t& v_n) -> bool {
            uint32_t t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= n_units) return false;
            v_n = vertex_of(t);
            n_ = load_vertex_buf<MASK, kDynLoadAux>(vb, v_n);
            return true;
        };
        while (hasA || hasB) {   // wave-uniform
            if (hasA) { process(A, vA); hasA = refill(A, vA); }
            if (hasB) { process(B, vB); hasB = refill(B, vB); }
        }
    }
}

uint32_t lbs_batch_grid(uint32_t total_units, const LbsTuning& t) {
    if (t.dyn) {     // four resident workgroups of four waves per CU; a workgroup should have a unit per wave at least
        const uint32_t grid = (uint32_t)kCUs * (1024u / kBatchDynBlock), max_useful = (total_units + 3) / 4;
        return grid > max_useful ? max_useful : grid;
    }
    uint32_t grid = (uint32_t)kCUs * (uint32_t)(t.blocks_per_cu > 0 ? t.blocks_per_cu : 1);
    const uint32_t max_useful = (total_units + 7) / 8;
    return grid > max_useful ? max_useful : grid;
}

template <bool EXACT>
static hipError_t launch_batch_mask(const LbsSegDev* d_segs, uint32_t n_segs, const uint32_t* d_block_seg, uint32_t grid,
                                    uint32_t total_units, size_t lds, int mask, bool dyn, hipStream_t s) {
#define FYX_BATCH_CASE(M)                                                                                          \
    case M:                                                                                                        \
        if (dyn) hipLaunchKernelGGL((lbs_skin_batch_dyn<EXACT, M>), dim3(grid), dim3(kBatchDynBlock), lds, s, d_segs, n_segs, d_block_seg, total_units); \
        else hipLaunchKernelGGL((lbs_skin_batch<EXACT, M>), dim3(grid), dim3(512), lds, s, d_segs, n_segs, d_block_seg, \
                           total_units);                                                                           \
        break;
    switch (mask) {
        FYX_BATCH_CASE(1) FYX_BATCH_CASE(2) FYX_BATCH_CASE(3) FYX_BATCH_CASE(4) FYX_BATCH_CASE(5) FYX_BATCH_CASE(6) FYX_BATCH_CASE(7)
        default: return hipSuccess;
    }
#undef FYX_BATCH_CASE
    return hipGetLastError();
}

hipError_t launch_lbs_batch(const LbsSegDev* d_segs, uint32_t n_segs, const uint32_t* d_block_seg, uint32_t grid,
                            uint32_t total_units, uint32_t max_bones, int mask, const LbsTuning& t, hipStream_t stream) {
    if (n_segs == 0 || total_units == 0 || grid == 0) return hipSuccess;
    const size_t lds = (size_t)max_bones * 64 + 64;
    const bool dyn = t.dyn != 0 && max_bones <= 256;     // (the grid was made by lbs_batch_grid from the same tuning)
    return t.exact ? launch_batch_mask<true>(d_segs, n_segs, d_block_seg, grid, total_units, lds, mask, dyn, stream)
                   : launch_batch_mask<false>(d_segs, n_segs, d_block_seg, grid, total_units, lds, mask, dyn, stream);
}

// ---------------------------------------------------------------------------------------
// Extended skinning kernel: blend shapes before skinning and/or interleaved (AoS) output.
//
//   blend shapes (standard.shader:167-173): for i in 0..blendShapesCount:
//        p += offsets[i].position * w_i;  n += offsets[i].normal * w_i;  t.xyz += offsets[i].tangent * w_i
//     with the offsets texelFetch'ed from the RGB16F volume BlendShapesContainer::from_lists builds
//     (surface.rs:116-217; f16 -> f32 is exact) and w_i = BlendShape::weight / 100 (mesh/mod.rs:794-798).
//     On the device the volume is re-tiled once per upload to [shape][64-vertex tile][9 components][64 lanes]
//     f16, so each of the nine loads of a wave is one dense 128-byte span and a shape costs exactly its
//     18 B/vertex of HBM traffic.  The shape weights of an instance are wave-uniform (scalar loads).
//   interleaved output: position / normal / tangent.xyzw are stored straight into a vertex buffer with
//     the renderer's layout (stride and attribute offsets as VertexBuffer's layout gives them,
//     scene/mesh/buffer.rs:404-415), so the skinned mesh can be drawn (or read back) without a
//     re-interleave pass; bytes of other attributes (tex coords, ...) are left as the caller put them.
// Same work split as lbs_skin (persistent grid, contiguous unit ranges, palette staged per instance
// segment); EXACT keeps the unfused reference order (GLSL leaves contraction to the driver, the CPU
// restatement does not contract).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float h2f(uint16_t h) {
    return (float)__builtin_bit_cast(_Float16, h);
}

// slab_off (AOS only): 0 = every lane stores its own 12 / 16-byte pieces at the output stride (a wave's store touches 64 lines, a
// third of each); otherwise the byte offset in LDS of the waves' slabs of 64 x out_stride bytes: a unit's outputs are laid out there as
// they lie in the vertex buffer and streamed out as consecutive dwords -- every store instruction covers two or three whole lines --
// with the dwords that belong to no written attribute masked off (they are left alone in memory, as before).  Round 6: 1 M vertices
// into an AnimatedVertex buffer (SoA streams in) 76.9 -> 33.8 us, into a StaticVertex buffer 48.4 -> 23.9, with four blend shapes 85.3 -> 46.2.
template <bool EXACT, bool SHAPES, bool AOS>
__global__ __launch_bounds__(512) void lbs_skin_ex(LbsExArgs x, uint32_t units_per_inst, uint32_t total_units, uint32_t slab_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LbsArgs& a = x.a;
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    f32x4* row3 = rows + 3 * a.n_bones;
    uint32_t* wave_flag = reinterpret_cast<uint32_t*>(row3 + a.n_bones);
    constexpr uint32_t WPB = 512 / 64;
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    if (u_begin >= u_end) return;
    const uint32_t inst_first = u_begin / units_per_inst, inst_last = (u_end - 1) / units_per_inst;
    const bool has_n = a.nrm != nullptr, has_t = a.tan != nullptr;   // kernel-uniform
    // the staged form's kernel-uniform facts: dwords per vertex, which of them are written, how the dword-in-vertex index of a lane moves
    // from one 64-dword store to the next
    const uint32_t spd = AOS ? x.out_stride / 4u : 1u;
    uint64_t wmask = 0;
    if constexpr (AOS) {
        if (x.off_pos >= 0) wmask |= 7ull << (x.off_pos / 4);
        if (has_n && x.off_nrm >= 0) wmask |= 7ull << (x.off_nrm / 4);
        if (has_t && x.off_tan >= 0) wmask |= 15ull << (x.off_tan / 4);
    }
    const uint32_t f_step = 64u % spd, f_first = lane % spd;

    for (uint32_t inst = inst_first; inst <= inst_last; ++inst) {
        const uint32_t inst_u0 = inst * units_per_inst;
        const uint32_t seg_b = (u_begin > inst_u0 ? u_begin : inst_u0) - inst_u0;
        const uint32_t seg_e = (u_end < inst_u0 + units_per_inst ? u_end : inst_u0 + units_per_inst) - inst_u0;
        const PaletteRegs pr = palette_fetch(a.palette + (size_t)inst * a.n_bones * 16, a.n_bones, tid);
        if (inst != inst_first) __syncthreads();
        const bool pj = palette_commit(pr, a.n_bones, rows, row3, tid);
        const bool wave_pj = __any(pj) != 0;
        if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;
        __syncthreads();
        bool projective = false;
#pragma unroll
        for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;
        const float* sw = SHAPES ? x.shape_w + (size_t)inst * x.n_shapes : nullptr;

        for (uint32_t u = seg_b + wave; u < seg_e; u += WPB) {  // wave-uniform
            const uint32_t v = u * 64 + lane;
            const uint32_t vs = v < a.n_verts ? v : 0;
            float px, py, pz, nx = 0.f, ny = 0.f, nz = 0.f;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            ld3<true>(a.pos + (size_t)vs * 3, px, py, pz);
            if (has_n) ld3<true>(a.nrm + (size_t)vs * 3, nx, ny, nz);
            if (has_t) t = ldg<true>(reinterpret_cast<const f32x4*>(a.tan) + vs);
            const f32x4 w = ldg<true>(reinterpret_cast<const f32x4*>(a.wgt) + vs);
            const uint32_t id = ldg<true>(a.idx + vs);
            if constexpr (SHAPES) {
                // tile u of every shape: 9 planes of 64 halfs; this lane's column
                const uint16_t* col = x.shapes + ((size_t)u * 9) * 64 + lane;
                const size_t shape_stride = (size_t)x.tiles_per_shape * 9 * 64;
#pragma unroll 2
                for (uint32_t sidx = 0; sidx < x.n_shapes; ++sidx) {
                    const uint16_t* c = col + (size_t)sidx * shape_stride;
                    const float ws = sw[sidx];
                    uint16_t h[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) h[k] = __builtin_nontemporal_load(c + k * 64);
                    if constexpr (EXACT) {
                        px = px + h2f(h[0]) * ws; py = py + h2f(h[1]) * ws; pz = pz + h2f(h[2]) * ws;
                        nx = nx + h2f(h[3]) * ws; ny = ny + h2f(h[4]) * ws; nz = nz + h2f(h[5]) * ws;
                        t.x = t.x + h2f(h[6]) * ws; t.y = t.y + h2f(h[7]) * ws; t.z = t.z + h2f(h[8]) * ws;
                    } else {
                        px = __builtin_fmaf(h2f(h[0]), ws, px); py = __builtin_fmaf(h2f(h[1]), ws, py);
                        pz = __builtin_fmaf(h2f(h[2]), ws, pz); nx = __builtin_fmaf(h2f(h[3]), ws, nx);
                        ny = __builtin_fmaf(h2f(h[4]), ws, ny); nz = __builtin_fmaf(h2f(h[5]), ws, nz);
                        t.x = __builtin_fmaf(h2f(h[6]), ws, t.x); t.y = __builtin_fmaf(h2f(h[7]), ws, t.y);
                        t.z = __builtin_fmaf(h2f(h[8]), ws, t.z);
                    }
                }
            }
            const Skinned o = skin_vertex<EXACT, 7>(rows, row3, projective, id, w, px, py, pz, nx, ny, nz, t.x, t.y, t.z);
            if constexpr (AOS) {
                if (slab_off) {      // kernel-uniform
                    unsigned char* slab = smem + slab_off + (size_t)wave * 64 * x.out_stride;
                    unsigned char* rec = slab + (size_t)lane * x.out_stride;       // (stride / 4 is odd for the engine's layouts: no bank conflicts)
                    if (x.off_pos >= 0) { float* q = reinterpret_cast<float*>(rec + x.off_pos); q[0] = o.px; q[1] = o.py; q[2] = o.pz; }
                    if (has_n && x.off_nrm >= 0) { float* q = reinterpret_cast<float*>(rec + x.off_nrm); q[0] = o.nx; q[1] = o.ny; q[2] = o.nz; }
                    if (has_t && x.off_tan >= 0) { float* q = reinterpret_cast<float*>(rec + x.off_tan); q[0] = o.tx; q[1] = o.ty; q[2] = o.tz; q[3] = t.w; }
                    __builtin_amdgcn_wave_barrier();      // a slab belongs to one wave, whose LDS operations complete in order
                    const uint32_t n_valid = (a.n_verts - u * 64) < 64u ? (a.n_verts - u * 64) : 64u;
                    const uint32_t n_dw = n_valid * spd;
                    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(slab);
                    uint32_t* o32 = reinterpret_cast<uint32_t*>(x.out_aos + ((size_t)inst * a.n_verts + (size_t)u * 64) * x.out_stride);
                    uint32_t f = f_first;
                    for (uint32_t d = lane; d < n_dw; d += 64) {
                        if ((wmask >> f) & 1ull) o32[d] = s32[d];      // (cacheable: nt stores measured the same at 68 bytes and 10 % slower at 48)
                        f += f_step;
                        if (f >= spd) f -= spd;
                    }
                    __builtin_amdgcn_wave_barrier();      // the slab is rewritten by the wave's next unit
                    continue;
                }
            }
            if (v < a.n_verts) {
                const size_t ov = (size_t)inst * a.n_verts + v;
                if constexpr (AOS) {
                    unsigned char* rec = x.out_aos + ov * x.out_stride;
                    if (x.off_pos >= 0) st3<true>(reinterpret_cast<float*>(rec + x.off_pos), o.px, o.py, o.pz);
                    if (has_n && x.off_nrm >= 0) st3<true>(reinterpret_cast<float*>(rec + x.off_nrm), o.nx, o.ny, o.nz);
                    if (has_t && x.off_tan >= 0) {
                        float* tp = reinterpret_cast<float*>(rec + x.off_tan);
                        st3<true>(tp, o.tx, o.ty, o.tz);
                        stg<true>(tp + 3, t.w);
                    }
                } else {
                    if (a.out_pos) st3<true>(a.out_pos + ov * 3, o.px, o.py, o.pz);
                    if (a.out_nrm) st3<true>(a.out_nrm + ov * 3, o.nx, o.ny, o.nz);
                    if (a.out_tan) stg<true>(reinterpret_cast<f32x4*>(a.out_tan) + ov, f32x4{o.tx, o.ty, o.tz, t.w});
                }
            }
        }
    }
}

template <bool EXACT, bool SHAPES, bool AOS>
static hipError_t launch_ex_one(const LbsExArgs& x, const LbsTuning& t, hipStream_t s) {
    const uint32_t upi = (x.a.n_verts + 63) / 64;
    const uint64_t total64 = (uint64_t)upi * x.a.n_instances;
    if (total64 == 0) return hipSuccess;
    if (total64 > 0xffffffffull) return hipErrorInvalidValue;
    const uint32_t total = (uint32_t)total64;
    // 88-100 VGPRs: 4-5 waves per SIMD = two resident 512-thread workgroups per CU; a persistent grid
    // larger than what is resident would run in rounds
    (void)t;
    uint32_t grid = (uint32_t)kCUs * 2u;
    const uint32_t max_useful = (total + 7) / 8;
    if (grid > max_useful) grid = max_useful;
    size_t lds = (size_t)x.a.n_bones * 64 + 64;
    uint32_t slab_off = 0;
    if (AOS && x.out_stride >= 4 && x.out_stride <= 160) {      // (<= 40 dwords per vertex: the written-dword mask is one 64-bit word; 8 slabs <= 80 KB)
        slab_off = (uint32_t)lds;
        lds += (size_t)8 * 64 * x.out_stride;
    }
    hipLaunchKernelGGL((lbs_skin_ex<EXACT, SHAPES, AOS>), dim3(grid), dim3(512), lds, s, x, upi, total, slab_off);
    return hipGetLastError();
}

hipError_t launch_lbs_ex(const LbsExArgs& x, const LbsTuning& t, hipStream_t s) {
    const bool shapes = x.n_shapes > 0, aos = x.out_aos != nullptr;
    const int key = (t.exact ? 4 : 0) | (shapes ? 2 : 0) | (aos ? 1 : 0);
    switch (key) {
        case 7: return launch_ex_one<true, true, true>(x, t, s);
        case 6: return launch_ex_one<true, true, false>(x, t, s);
        case 5: return launch_ex_one<true, false, true>(x, t, s);
        case 4: return launch_ex_one<true, false, false>(x, t, s);
        case 3: return launch_ex_one<false, true, true>(x, t, s);
        case 2: return launch_ex_one<false, true, false>(x, t, s);
        case 1: return launch_ex_one<false, false, true>(x, t, s);
        default: return launch_ex_one<false, false, false>(x, t, s);
    }
}

// ---------------------------------------------------------------------------------------
// Vertex-buffer-in, vertex-buffer-out skinning (the engine's native interleaved layout on both sides).
//
// Scattered 12/16-byte stores at a 68-byte stride only fill parts of every 128-byte line and run at ~1.3 TB/s,
// so this kernel moves whole spans instead: a wave's unit of 64 vertices is ONE contiguous 64*stride-byte span
// of the VertexBuffer (4352 B for AnimatedVertex).  The wave
//   1. loads the span with dense 16-byte lane accesses (the NEXT unit's span is already in registers while the
//      current one is processed) and parks it in its private LDS slab,
//   2. every lane reads the fields of its own vertex from LDS (stride/4 is odd for the engine's layouts, so the
//      64 lanes hit different banks), applies blend-shape offsets, skins, and writes position / normal /
//      tangent.xyz back in place,
//   3. streams the slab out again with dense 16-byte stores.
// HBM traffic is 2 * stride bytes per vertex (136 B for AnimatedVertex: texture coordinates, bone weights and
// indices pass through, so the output is a complete vertex buffer for the renderer's geometry cache,
// renderer/cache/geometry.rs:84-93).  No barrier besides the palette staging: a slab belongs to one wave and LDS
// operations of a wave complete in order.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kAosMaxF4PerLane = 10;   // stride <= 160 bytes
constexpr int kAosBlock = 256;

// PL = 16-byte accesses per lane and span = ceil(stride / 16): a compile-time bound so that only the registers a
// layout needs are held for the span in flight (5 for the 68-byte AnimatedVertex)
// One segment (= the units [seg_b, seg_e) of one instance of one mesh) of the whole-span kernel: shared by the
// single-mesh launch (segments = the instances of x) and the batched one (segments = the batch's table).
// palette / sw / out_inst: this instance's matrices, blend-shape weights and output vertex buffer.
template <bool EXACT, bool SHAPES, uint32_t PL>
__device__ __forceinline__ void aos_segment(const LbsExArgs& x, const float* __restrict__ palette, const float* __restrict__ sw,
                                            unsigned char* __restrict__ out_inst, uint32_t seg_b, uint32_t seg_e, bool first,
                                            unsigned char* smem) {
    const LbsArgs& a = x.a;
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    f32x4* row3 = rows + 3 * a.n_bones;
    uint32_t* wave_flag = reinterpret_cast<uint32_t*>(row3 + a.n_bones);   // 16 words: [0, WPB) the waves' projective flags, [8] the unit ticket
    uint32_t* ticket = wave_flag + 8;
    constexpr uint32_t WPB = kAosBlock / 64;
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t stride = x.out_stride;                     // == input stride
    const uint32_t span_bytes = 64u * stride;
    unsigned char* slab = smem + (size_t)a.n_bones * 64 + 64 + (size_t)wave * 64 * stride;
    f32x4* slab4 = reinterpret_cast<f32x4*>(slab);
    const PaletteRegs pr = palette_fetch(palette, a.n_bones, tid);
    // Round 5 (what paid for lbs_skin_dyn): the waves DRAW their units from an LDS ticket (a wave that is served faster
    // takes more; the segment ends when its work does), TWO spans are in flight per wave (a register set is refilled as soon as its
    // span is parked in the slab, while the other set's loads have had a whole unit's time to land), the spans are buffer resources
    // of exactly one span (a lane past the span's last 16 bytes loads zeros and stores nothing: no per-lane bounds tests), loads `nt`,
    // stores `sc1`.  The input buffer is padded by one unit, so a ragged last span may be read in full.
    auto span_in = [&](uint32_t u) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(x.in_aos) + (size_t)u * span_bytes, 0, span_bytes, 0x00020000); };
    auto load_span = [&](f32x4 (&r)[PL], uint32_t u) {
        const __amdgpu_buffer_rsrc_t rs = span_in(u);
#pragma unroll
        for (uint32_t k = 0; k < PL; ++k) r[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + 64u * k) * 16u, 0, 2));
    };
    // (with blend shapes the second register set costs the fourth wave per SIMD and the launch gets slower -- 42.4 against 40.3 us
    // alone, measured: profiles/r05_vertex_buffer.jsonl -- so that form keeps ONE span in flight: ticket, buffer spans and sc1 stores only)
    // (and the wide layouts, 32 / 40 bytes of span per lane: two sets of those would leave two waves per SIMD)
    constexpr bool TWO = !SHAPES && PL <= 5;
    const uint32_t n_units = seg_e - seg_b;
    uint32_t uA = seg_b + wave, uB = uA + WPB;
    bool vA = wave < n_units, vB = TWO && WPB + wave < n_units;
    f32x4 A[PL], B[TWO ? PL : 1];
    if (vA) load_span(A, uA);
    if (!first) __syncthreads();   // every wave is done with the previous palette (and the previous segment's ticket)
    if (tid == 0) *ticket = (TWO ? 2 : 1) * WPB;
    const bool pj = palette_commit(pr, a.n_bones, rows, row3, tid);
    const bool wave_pj = __any(pj) != 0;
    if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;
    __syncthreads();
    bool projective = false;
#pragma unroll
    for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;
    if constexpr (TWO) if (vB) load_span(B, uB);     // (behind the staging barrier, as lbs_skin_dyn: the palette columns of the later waves do not queue behind it)

    // one unit: park the span in the slab, refill the register set, skin in the slab, stream the slab out
    auto process = [&](f32x4 (&r)[PL], uint32_t& u, bool& valid) {
#pragma unroll
        for (uint32_t k = 0; k < PL; ++k) {
            const uint32_t i = lane + 64u * k;
            if (i * 16u < span_bytes) slab4[i] = r[k];
        }
        const uint32_t u_now = u;
        {
            uint32_t t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = __builtin_amdgcn_readfirstlane(t);
            valid = t < n_units;
            u = seg_b + t;
            if (valid) load_span(r, u);
        }
        __builtin_amdgcn_wave_barrier();
        unsigned char* rec = slab + (size_t)lane * stride;
        float* pp = reinterpret_cast<float*>(rec + x.off_pos);
        float px = pp[0], py = pp[1], pz = pp[2];
        float nx = 0.f, ny = 0.f, nz = 0.f;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        float* np_ = x.off_nrm >= 0 ? reinterpret_cast<float*>(rec + x.off_nrm) : nullptr;
        float* tp = x.off_tan >= 0 ? reinterpret_cast<float*>(rec + x.off_tan) : nullptr;
        if (np_) { nx = np_[0]; ny = np_[1]; nz = np_[2]; }
        if (tp) { t.x = tp[0]; t.y = tp[1]; t.z = tp[2]; }
        const float* wp = reinterpret_cast<const float*>(rec + x.in_off_wgt);
        const f32x4 w = {wp[0], wp[1], wp[2], wp[3]};
        const uint32_t id = *reinterpret_cast<const uint32_t*>(rec + x.in_off_idx);
        if constexpr (SHAPES) {
            const uint16_t* col = x.shapes + ((size_t)u_now * 9) * 64 + lane;
            const size_t shape_stride = (size_t)x.tiles_per_shape * 9 * 64;
#pragma unroll 1
            for (uint32_t sidx = 0; sidx < x.n_shapes; ++sidx) {
                const uint16_t* c = col + (size_t)sidx * shape_stride;
                const float ws = sw[sidx];
                uint16_t h[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) h[k] = __builtin_nontemporal_load(c + k * 64);
                if constexpr (EXACT) {
                    px = px + h2f(h[0]) * ws; py = py + h2f(h[1]) * ws; pz = pz + h2f(h[2]) * ws;
                    nx = nx + h2f(h[3]) * ws; ny = ny + h2f(h[4]) * ws; nz = nz + h2f(h[5]) * ws;
                    t.x = t.x + h2f(h[6]) * ws; t.y = t.y + h2f(h[7]) * ws; t.z = t.z + h2f(h[8]) * ws;
                } else {
                    px = __builtin_fmaf(h2f(h[0]), ws, px); py = __builtin_fmaf(h2f(h[1]), ws, py);
                    pz = __builtin_fmaf(h2f(h[2]), ws, pz); nx = __builtin_fmaf(h2f(h[3]), ws, nx);
                    ny = __builtin_fmaf(h2f(h[4]), ws, ny); nz = __builtin_fmaf(h2f(h[5]), ws, nz);
                    t.x = __builtin_fmaf(h2f(h[6]), ws, t.x); t.y = __builtin_fmaf(h2f(h[7]), ws, t.y);
                    t.z = __builtin_fmaf(h2f(h[8]), ws, t.z);
                }
            }
        }
        const Skinned o = skin_vertex<EXACT, 7>(rows, row3, projective, id, w, px, py, pz, nx, ny, nz, t.x, t.y, t.z);
        pp[0] = o.px; pp[1] = o.py; pp[2] = o.pz;
        if (np_) { np_[0] = o.nx; np_[1] = o.ny; np_[2] = o.nz; }
        if (tp) { tp[0] = o.tx; tp[1] = o.ty; tp[2] = o.tz; }
        __builtin_amdgcn_wave_barrier();
        // stream the slab out; only whole vertices of a ragged last unit
        const uint32_t n_valid = (a.n_verts - u_now * 64) < 64u ? (a.n_verts - u_now * 64) : 64u;
        unsigned char* out_span = out_inst + (size_t)u_now * span_bytes;
        if (n_valid == 64u && ((reinterpret_cast<uintptr_t>(out_span) & 15u) == 0)) {
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out_span, 0, span_bytes, 0x00020000);
#pragma unroll
            for (uint32_t k = 0; k < PL; ++k) {
                const uint32_t i = lane + 64u * k;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (i * 16u < span_bytes) v = slab4[i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, i * 16u, 0, 16);
            }
        } else {
            const uint32_t n_dw = n_valid * (stride / 4);
            const uint32_t* s32 = reinterpret_cast<const uint32_t*>(slab);
            uint32_t* o32 = reinterpret_cast<uint32_t*>(out_span);
            for (uint32_t i = lane; i < n_dw; i += 64) o32[i] = s32[i];
        }
        __builtin_amdgcn_wave_barrier();   // the slab is rewritten by the next unit
    };
    for (;;) {   // wave-uniform
        if (!vA) break;
        process(A, uA, vA);
        if constexpr (TWO) {
            if (!vB) break;
            process(B, uB, vB);
        }
    }
}

template <bool EXACT, bool SHAPES, uint32_t PL>
__global__ __launch_bounds__(kAosBlock, (SHAPES && PL <= 5) ? 4 : 1) void lbs_skin_aos(LbsExArgs x, uint32_t units_per_inst, uint32_t total_units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LbsArgs& a = x.a;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    if (u_begin >= u_end) return;
    const uint32_t inst_first = u_begin / units_per_inst, inst_last = (u_end - 1) / units_per_inst;
    for (uint32_t inst = inst_first; inst <= inst_last; ++inst) {
        const uint32_t inst_u0 = inst * units_per_inst;
        const uint32_t seg_b = (u_begin > inst_u0 ? u_begin : inst_u0) - inst_u0;
        const uint32_t seg_e = (u_end < inst_u0 + units_per_inst ? u_end : inst_u0 + units_per_inst) - inst_u0;
        aos_segment<EXACT, SHAPES, PL>(x, a.palette + (size_t)inst * a.n_bones * 16,
                                       SHAPES ? x.shape_w + (size_t)inst * x.n_shapes : nullptr,
                                       x.out_aos + (size_t)inst * a.n_verts * x.out_stride, seg_b, seg_e, inst == inst_first, smem);
    }
}

// Batched form (fyx_lbs_skin_ex_batch): the segments of many meshes in one unit numbering, as lbs_skin_batch.
template <bool EXACT, bool SHAPES, uint32_t PL>
__global__ __launch_bounds__(kAosBlock, (SHAPES && PL <= 5) ? 4 : 1) void lbs_skin_aos_batch(const LbsExSegDev* __restrict__ segs_g, uint32_t n_segs,
                                                               const uint32_t* __restrict__ block_seg, uint32_t total_units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FYX_CONSTANT LbsExSegDev* segs = (const FYX_CONSTANT LbsExSegDev*)segs_g;
    const uint32_t u_begin = (uint32_t)(((uint64_t)blockIdx.x * total_units) / gridDim.x);
    const uint32_t u_end = (uint32_t)(((uint64_t)(blockIdx.x + 1) * total_units) / gridDim.x);
    if (u_begin >= u_end) return;
    bool first = true;
    for (uint32_t seg = block_seg[blockIdx.x]; seg < n_segs; ++seg) {   // workgroup-uniform
        const uint32_t s_u0 = segs[seg].unit0;
        if (s_u0 >= u_end) break;
        LbsExArgs x;
        x.a.n_verts = segs[seg].n_verts; x.a.n_bones = segs[seg].n_bones;
        x.in_aos = segs[seg].in_aos;
        x.shapes = segs[seg].shapes; x.n_shapes = segs[seg].n_shapes; x.tiles_per_shape = segs[seg].tiles_per_shape;
        x.out_stride = segs[seg].stride;
        x.off_pos = segs[seg].off_pos; x.off_nrm = segs[seg].off_nrm; x.off_tan = segs[seg].off_tan;
        x.in_off_wgt = segs[seg].in_off_wgt; x.in_off_idx = segs[seg].in_off_idx;
        const uint32_t upi = (x.a.n_verts + 63) / 64;
        if (s_u0 + upi <= u_begin) continue;
        const uint32_t seg_b = (u_begin > s_u0 ? u_begin : s_u0) - s_u0;
        const uint32_t seg_e = (u_end < s_u0 + upi ? u_end : s_u0 + upi) - s_u0;
        aos_segment<EXACT, SHAPES, PL>(x, segs[seg].palette, segs[seg].shape_w, segs[seg].out_aos, seg_b, seg_e, first, smem);
        first = false;
    }
}

// persistent grid = exactly what is resident (registers and LDS decide); asked once per (kernel, LDS size), not per
// launch (per host thread: a fyx_ctx is single-threaded; every context runs the same code object)
template <typename K>
static hipError_t aos_blocks_per_cu(K kernel, size_t lds, int* per_cu) {
    struct Entry { const void* fn; size_t lds; int per_cu; };
    constexpr int kEntries = 16;
    static thread_local Entry cache[kEntries] = {};
    static thread_local int next = 0;
    const void* fn = reinterpret_cast<const void*>(kernel);
    for (const Entry& e : cache)
        if (e.fn == fn && e.lds == lds) { *per_cu = e.per_cu; return hipSuccess; }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int q = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, fn, kAosBlock, lds);
    if (e != hipSuccess) return e;
    cache[next] = Entry{fn, lds, q < 1 ? 1 : q};
    next = (next + 1) % kEntries;
    *per_cu = q < 1 ? 1 : q;
    return hipSuccess;
}

static size_t aos_lds_bytes(uint32_t n_bones, uint32_t stride) {
    return (size_t)n_bones * 64 + 64 + (size_t)(kAosBlock / 64) * 64 * stride;
}

template <bool EXACT, bool SHAPES, uint32_t PL>
static hipError_t launch_aos_pl(const LbsExArgs& x, hipStream_t s) {
    const uint32_t upi = (x.a.n_verts + 63) / 64;
    const uint64_t total64 = (uint64_t)upi * x.a.n_instances;
    if (total64 == 0) return hipSuccess;
    if (total64 > 0xffffffffull) return hipErrorInvalidValue;
    const uint32_t total = (uint32_t)total64;
    constexpr uint32_t WPB = kAosBlock / 64;
    const size_t lds = aos_lds_bytes(x.a.n_bones, x.out_stride);
    int per_cu = 1;
    if (hipError_t e = aos_blocks_per_cu(&lbs_skin_aos<EXACT, SHAPES, PL>, lds, &per_cu); e != hipSuccess) return e;
    uint32_t grid = (uint32_t)kCUs * (uint32_t)per_cu;
    const uint32_t max_useful = (total + WPB - 1) / WPB;
    if (grid > max_useful) grid = max_useful;
    hipLaunchKernelGGL((lbs_skin_aos<EXACT, SHAPES, PL>), dim3(grid), dim3(kAosBlock), lds, s, x, upi, total);
    return hipGetLastError();
}

uint32_t lbs_aos_bucket(uint32_t stride) {
    if (stride == 0 || (stride & 3u) || stride * 4 > 64 * kAosMaxF4PerLane) return 0;
    const uint32_t pl = (stride * 4 + 63) / 64;
    return pl <= 4 ? 4 : pl <= 5 ? 5 : pl <= 8 ? 8 : kAosMaxF4PerLane;
}

template <bool EXACT, bool SHAPES>
static hipError_t launch_aos_one(const LbsExArgs& x, hipStream_t s) {
    switch (lbs_aos_bucket(x.out_stride)) {
        case 4: return launch_aos_pl<EXACT, SHAPES, 4>(x, s);
        case 5: return launch_aos_pl<EXACT, SHAPES, 5>(x, s);
        case 8: return launch_aos_pl<EXACT, SHAPES, 8>(x, s);
        default: return launch_aos_pl<EXACT, SHAPES, kAosMaxF4PerLane>(x, s);
    }
}

hipError_t launch_lbs_aos(const LbsExArgs& x, const LbsTuning& t, hipStream_t s) {
    if (!lbs_aos_bucket(x.out_stride)) return hipErrorInvalidValue;
    const bool shapes = x.n_shapes > 0;
    if (t.exact) return shapes ? launch_aos_one<true, true>(x, s) : launch_aos_one<true, false>(x, s);
    return shapes ? launch_aos_one<false, true>(x, s) : launch_aos_one<false, false>(x, s);
}

// The batched launches: `op` is "how many workgroups per CU are resident" (grid != nullptr) or "launch".
template <bool EXACT, bool SHAPES, uint32_t PL>
static hipError_t aos_batch_pl(size_t lds, int* per_cu, const LbsExSegDev* d_segs, uint32_t n_segs, const uint32_t* d_block_seg,
                               uint32_t grid, uint32_t total_units, hipStream_t s) {
    if (per_cu) return aos_blocks_per_cu(&lbs_skin_aos_batch<EXACT, SHAPES, PL>, lds, per_cu);
    hipLaunchKernelGGL((lbs_skin_aos_batch<EXACT, SHAPES, PL>), dim3(grid), dim3(kAosBlock), lds, s, d_segs, n_segs, d_block_seg,
                       total_units);
    return hipGetLastError();
}

template <bool EXACT, bool SHAPES>
static hipError_t aos_batch_bucket(uint32_t bucket, size_t lds, int* per_cu, const LbsExSegDev* d_segs, uint32_t n_segs,
                                   const uint32_t* d_block_seg, uint32_t grid, uint32_t total_units, hipStream_t s) {
    switch (bucket) {
        case 4: return aos_batch_pl<EXACT, SHAPES, 4>(lds, per_cu, d_segs, n_segs, d_block_seg, grid, total_units, s);
        case 5: return aos_batch_pl<EXACT, SHAPES, 5>(lds, per_cu, d_segs, n_segs, d_block_seg, grid, total_units, s);
        case 8: return aos_batch_pl<EXACT, SHAPES, 8>(lds, per_cu, d_segs, n_segs, d_block_seg, grid, total_units, s);
        case kAosMaxF4PerLane: return aos_batch_pl<EXACT, SHAPES, kAosMaxF4PerLane>(lds, per_cu, d_segs, n_segs, d_block_seg, grid, total_units, s);
        default: return hipErrorInvalidValue;
    }
}

static hipError_t aos_batch_any(bool exact, bool shapes, uint32_t bucket, size_t lds, int* per_cu, const LbsExSegDev* d_segs,
                                uint32_t n_segs, const uint32_t* d_block_seg, uint32_t grid, uint32_t total_units, hipStream_t s) {
    if (exact) return shapes ? aos_batch_bucket<true, true>(bucket, lds, per_cu, d_segs, n_segs, d_block_seg, grid, total_units, s)
                             : aos_batch_bucket<true, false>(bucket, lds, per_cu, d_segs, n_segs, d_block_seg, grid, total_units, s);
    return shapes ? aos_batch_bucket<false, true>(bucket, lds, per_cu, d_segs, n_segs, d_block_seg, grid, total_units, s)
                  : aos_batch_bucket<false, false>(bucket, lds, per_cu, d_segs, n_segs, d_block_seg, grid, total_units, s);
}

hipError_t lbs_aos_batch_grid(uint32_t total_units, uint32_t max_bones, uint32_t max_stride, uint32_t bucket, bool shapes,
                              const LbsTuning& t, uint32_t* grid) {
    int per_cu = 1;
    if (hipError_t e = aos_batch_any(t.exact != 0, shapes, bucket, aos_lds_bytes(max_bones, max_stride), &per_cu, nullptr, 0,
                                     nullptr, 0, 0, nullptr); e != hipSuccess) return e;
    uint32_t g = (uint32_t)kCUs * (uint32_t)per_cu;
    const uint32_t max_useful = (total_units + (kAosBlock / 64) - 1) / (kAosBlock / 64);
    *grid = g > max_useful ? max_useful : g;
    return hipSuccess;
}

hipError_t launch_lbs_aos_batch(const LbsExSegDev* d_segs, uint32_t n_segs, const uint32_t* d_block_seg, uint32_t grid,
                                uint32_t total_units, uint32_t max_bones, uint32_t max_stride, uint32_t bucket, bool shapes,
                                const LbsTuning& t, hipStream_t stream) {
    if (n_segs == 0 || total_units == 0 || grid == 0) return hipSuccess;
    return aos_batch_any(t.exact != 0, shapes, bucket, aos_lds_bytes(max_bones, max_stride), nullptr, d_segs, n_segs, d_block_seg,
                         grid, total_units, stream);
}

// RGB16F volume (engine layout: [shape][vertex][texel: position, normal, tangent][rgb], 18 B per vertex,
// plane_vertices = width * height texel triples per shape) -> [shape][tile][9][64] f16.
__global__ __launch_bounds__(256) void retile_blend_shapes_kernel(const uint16_t* __restrict__ src, uint32_t n_verts,
                                                                  uint32_t plane_vertices, uint32_t n_shapes,
                                                                  uint32_t tiles, uint16_t* __restrict__ dst) {
    const uint64_t total = (uint64_t)n_shapes * tiles * 9 * 64;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = (uint32_t)(e & 63), k = (uint32_t)((e >> 6) % 9);
        const uint64_t ts = (e >> 6) / 9;
        const uint32_t tile = (uint32_t)(ts % tiles), shape = (uint32_t)(ts / tiles);
        const uint32_t v = tile * 64 + l;
        dst[e] = v < n_verts ? src[((size_t)shape * plane_vertices + v) * 9 + k] : (uint16_t)0;
    }
}

hipError_t launch_retile_blend_shapes(const uint16_t* d_src, uint32_t n_verts, uint32_t plane_vertices,
                                      uint32_t n_shapes, uint16_t* d_dst, hipStream_t s) {
    const uint32_t tiles = (n_verts + 63) / 64;
    const uint64_t total = (uint64_t)n_shapes * tiles * 9 * 64;
    if (total == 0) return hipSuccess;
    uint64_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(retile_blend_shapes_kernel, dim3((uint32_t)grid), dim3(256), 0, s, d_src, n_verts,
                       plane_vertices, n_shapes, tiles, d_dst);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// AoS -> SoA de-interleave.  One thread per vertex; field reads are 4-byte loads at the
// vertex stride (the AoS source is read once per mesh modification, not per frame).
// Bytes are reinterpreted as little-endian f32/u8 exactly as VertexReadTrait does
// (fyrox-impl/src/scene/mesh/buffer.rs:1279-1321).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rd_u32_unaligned(const uint8_t* p) {
    if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) return *reinterpret_cast<const uint32_t*>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

__global__ __launch_bounds__(256) void deinterleave_kernel(
    const uint8_t* __restrict__ aos, uint32_t n_verts, uint32_t stride, int off_pos, int off_nrm,
    int off_tan, int off_wgt, int off_idx, uint32_t* __restrict__ pos, uint32_t* __restrict__ nrm,
    uint32_t* __restrict__ tan, uint32_t* __restrict__ wgt, uint32_t* __restrict__ idx) {
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n_verts;
         v += gridDim.x * blockDim.x) {
        const uint8_t* b = aos + (size_t)v * stride;
        for (int i = 0; i < 3; ++i) pos[(size_t)v * 3 + i] = rd_u32_unaligned(b + off_pos + 4 * i);
        if (off_nrm >= 0)
            for (int i = 0; i < 3; ++i) nrm[(size_t)v * 3 + i] = rd_u32_unaligned(b + off_nrm + 4 * i);
        if (off_tan >= 0)
            for (int i = 0; i < 4; ++i) tan[(size_t)v * 4 + i] = rd_u32_unaligned(b + off_tan + 4 * i);
        for (int i = 0; i < 4; ++i) wgt[(size_t)v * 4 + i] = rd_u32_unaligned(b + off_wgt + 4 * i);
        idx[v] = rd_u32_unaligned(b + off_idx);
    }
}

hipError_t launch_deinterleave(const uint8_t* d_aos, uint32_t n_verts, uint32_t stride, int off_pos,
                               int off_nrm, int off_tan, int off_wgt, int off_idx, float* d_pos,
                               float* d_nrm, float* d_tan, float* d_wgt, uint32_t* d_idx,
                               hipStream_t stream) {
    if (n_verts == 0) return hipSuccess;
    uint32_t grid = (n_verts + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(deinterleave_kernel, dim3(grid), dim3(256), 0, stream, d_aos, n_verts, stride,
                       off_pos, off_nrm, off_tan, off_wgt, off_idx,
                       reinterpret_cast<uint32_t*>(d_pos), reinterpret_cast<uint32_t*>(d_nrm),
                       reinterpret_cast<uint32_t*>(d_tan), reinterpret_cast<uint32_t*>(d_wgt), d_idx);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// max bone index (validation at upload; skinning rejects palettes shorter than max+1,
// where the Rust loop would panic on `bone_matrices[bone_index as usize]`).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void max_bone_index_kernel(const uint32_t* __restrict__ idx,
                                                             uint32_t n, uint32_t* out) {
    __shared__ uint32_t sm[4];
    uint32_t m = 0;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        const uint32_t id = idx[v];
        m = max(m, max(max(id & 0xffu, (id >> 8) & 0xffu), max((id >> 16) & 0xffu, id >> 24)));
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(sm[0], sm[1]), max(sm[2], sm[3])));  // one per block
}

hipError_t launch_max_bone_index(const uint32_t* d_idx, uint32_t n_verts, uint32_t* d_out,
                                 hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess || n_verts == 0) return e;
    uint32_t grid = (n_verts + 255) / 256;
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(max_bone_index_kernel, dim3(grid), dim3(256), 0, stream, d_idx, n_verts, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// AABB of skinned positions (Mesh::accurate_world_bounding_box, scene/mesh/mod.rs:470-526;
// default box = (+MAX, -MAX), add_point = strict </> compares, fyrox-math/src/aabb.rs:33-104).
// min/max are exact, so any reduction order gives the reference's result.
// ---------------------------------------------------------------------------------------
constexpr int kAabbBlock = 256;
constexpr uint32_t kAabbMaxBlocks = 2048;

uint32_t aabb_partial_blocks(uint32_t n) {
    uint32_t g = (n + kAabbBlock - 1) / kAabbBlock;
    if (g == 0) g = 1;
    return g > kAabbMaxBlocks ? kAabbMaxBlocks : g;
}

__device__ __forceinline__ void block_minmax_store(float mn[3], float mx[3], float* partial) {
    __shared__ float sm[kAabbBlock / 64][6];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        for (int o = 32; o > 0; o >>= 1) {
            mn[i] = fminf(mn[i], __shfl_xor(mn[i], o, 64));
            mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o, 64));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 3; ++i) { sm[wave][i] = mn[i]; sm[wave][3 + i] = mx[i]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kAabbBlock / 64; ++w)
            for (int i = 0; i < 3; ++i) {
                sm[0][i] = fminf(sm[0][i], sm[w][i]);
                sm[0][3 + i] = fmaxf(sm[0][3 + i], sm[w][3 + i]);
            }
        for (int i = 0; i < 6; ++i) partial[i] = sm[0][i];
    }
}

__global__ __launch_bounds__(kAabbBlock) void skinned_aabb_kernel(LbsArgs a, float* partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    f32x4* row3 = rows + 3 * a.n_bones;
    const bool pj = stage_palette(a.palette, a.n_bones, rows, row3, threadIdx.x, kAabbBlock);
    const bool projective = __syncthreads_or(pj) != 0;
    float mn[3] = {__FLT_MAX__, __FLT_MAX__, __FLT_MAX__};
    float mx[3] = {-__FLT_MAX__, -__FLT_MAX__, -__FLT_MAX__};
    for (uint32_t v = blockIdx.x * kAabbBlock + threadIdx.x; v < a.n_verts;
         v += gridDim.x * kAabbBlock) {
        float px, py, pz;
        ld3<false>(a.pos + (size_t)v * 3, px, py, pz);
        const f32x4 w = reinterpret_cast<const f32x4*>(a.wgt)[v];
        const Skinned o = skin_vertex<true, 1>(rows, row3, projective, a.idx[v], w, px, py, pz, 0, 0,
                                               0, 0, 0, 0);
        // strict compares as add_point: NaN never replaces a bound
        if (o.px < mn[0]) mn[0] = o.px;
        if (o.py < mn[1]) mn[1] = o.py;
        if (o.pz < mn[2]) mn[2] = o.pz;
        if (o.px > mx[0]) mx[0] = o.px;
        if (o.py > mx[1]) mx[1] = o.py;
        if (o.pz > mx[2]) mx[2] = o.pz;
    }
    block_minmax_store(mn, mx, partials + (size_t)blockIdx.x * 6);
}

__global__ __launch_bounds__(kAabbBlock) void points_aabb_kernel(const float* __restrict__ xyz,
                                                                 uint64_t n, float* partials) {
    float mn[3] = {__FLT_MAX__, __FLT_MAX__, __FLT_MAX__};
    float mx[3] = {-__FLT_MAX__, -__FLT_MAX__, -__FLT_MAX__};
    for (uint64_t v = (uint64_t)blockIdx.x * kAabbBlock + threadIdx.x; v < n;
         v += (uint64_t)gridDim.x * kAabbBlock) {
        float x, y, z;
        ld3<false>(xyz + v * 3, x, y, z);
        if (x < mn[0]) mn[0] = x;
        if (y < mn[1]) mn[1] = y;
        if (z < mn[2]) mn[2] = z;
        if (x > mx[0]) mx[0] = x;
        if (y > mx[1]) mx[1] = y;
        if (z > mx[2]) mx[2] = z;
    }
    block_minmax_store(mn, mx, partials + (size_t)blockIdx.x * 6);
}

__global__ __launch_bounds__(kAabbBlock) void aabb_final_kernel(const float* partials, uint32_t n,
                                                                float* out) {
    float mn[3] = {__FLT_MAX__, __FLT_MAX__, __FLT_MAX__};
    float mx[3] = {-__FLT_MAX__, -__FLT_MAX__, -__FLT_MAX__};
    for (uint32_t b = threadIdx.x; b < n; b += kAabbBlock)
        for (int i = 0; i < 3; ++i) {
            mn[i] = fminf(mn[i], partials[(size_t)b * 6 + i]);
            mx[i] = fmaxf(mx[i], partials[(size_t)b * 6 + 3 + i]);
        }
    block_minmax_store(mn, mx, out);
}

// Instanced form: the box of EVERY instance of a crowd in one launch (palettes and boxes stay on the device).  Block
// (x, y) skins slice x of instance y's vertices with instance y's palette; with one slice per instance the block writes
// the instance's box itself, otherwise a second launch folds the slices.
__global__ __launch_bounds__(kAabbBlock) void skinned_aabb_inst_kernel(LbsArgs a, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* rows = reinterpret_cast<f32x4*>(smem);
    f32x4* row3 = rows + 3 * a.n_bones;
    const uint32_t inst = blockIdx.y;
    const bool pj = stage_palette(a.palette + (size_t)inst * a.n_bones * 16, a.n_bones, rows, row3, threadIdx.x, kAabbBlock);
    const bool projective = __syncthreads_or(pj) != 0;
    float mn[3] = {__FLT_MAX__, __FLT_MAX__, __FLT_MAX__};
    float mx[3] = {-__FLT_MAX__, -__FLT_MAX__, -__FLT_MAX__};
    for (uint32_t v = blockIdx.x * kAabbBlock + threadIdx.x; v < a.n_verts; v += gridDim.x * kAabbBlock) {
        float px, py, pz;
        ld3<false>(a.pos + (size_t)v * 3, px, py, pz);
        const f32x4 w = reinterpret_cast<const f32x4*>(a.wgt)[v];
        const Skinned o = skin_vertex<true, 1>(rows, row3, projective, a.idx[v], w, px, py, pz, 0, 0, 0, 0, 0, 0);
        if (o.px < mn[0]) mn[0] = o.px;
        if (o.py < mn[1]) mn[1] = o.py;
        if (o.pz < mn[2]) mn[2] = o.pz;
        if (o.px > mx[0]) mx[0] = o.px;
        if (o.py > mx[1]) mx[1] = o.py;
        if (o.pz > mx[2]) mx[2] = o.pz;
    }
    block_minmax_store(mn, mx, out + ((size_t)inst * gridDim.x + blockIdx.x) * 6);
}

__global__ __launch_bounds__(64) void aabb_final_inst_kernel(const float* partials, uint32_t slices, float* out) {
    const uint32_t inst = blockIdx.x;
    float mn[3] = {__FLT_MAX__, __FLT_MAX__, __FLT_MAX__};
    float mx[3] = {-__FLT_MAX__, -__FLT_MAX__, -__FLT_MAX__};
    for (uint32_t b = threadIdx.x; b < slices; b += 64)
        for (int i = 0; i < 3; ++i) {
            mn[i] = fminf(mn[i], partials[((size_t)inst * slices + b) * 6 + i]);
            mx[i] = fmaxf(mx[i], partials[((size_t)inst * slices + b) * 6 + 3 + i]);
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        for (int o = 32; o > 0; o >>= 1) {
            mn[i] = fminf(mn[i], __shfl_xor(mn[i], o, 64));
            mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o, 64));
        }
    if (threadIdx.x == 0)
        for (int i = 0; i < 3; ++i) { out[(size_t)inst * 6 + i] = mn[i]; out[(size_t)inst * 6 + 3 + i] = mx[i]; }
}

uint32_t aabb_inst_slices(uint32_t n_verts, uint32_t n_instances) {
    // enough blocks to fill the chip when the instances alone do not, at most 64 slices per instance
    uint32_t want = (uint32_t)((4ull * kCUs + n_instances - 1) / n_instances);
    const uint32_t cap = (n_verts + kAabbBlock - 1) / kAabbBlock;
    if (want > cap) want = cap;
    if (want > 64) want = 64;
    return want ? want : 1;
}

hipError_t launch_skinned_aabb_inst(const LbsArgs& a, float* d_partials, float* d_out, hipStream_t s) {
    if (a.n_instances == 0) return hipSuccess;
    if (a.n_instances > 65535u) return hipErrorInvalidValue;
    const uint32_t slices = aabb_inst_slices(a.n_verts, a.n_instances);
    hipLaunchKernelGGL(skinned_aabb_inst_kernel, dim3(slices, a.n_instances), dim3(kAabbBlock), (size_t)a.n_bones * 64, s, a,
                       slices == 1 ? d_out : d_partials);
    if (slices > 1) hipLaunchKernelGGL(aabb_final_inst_kernel, dim3(a.n_instances), dim3(64), 0, s, d_partials, slices, d_out);
    return hipGetLastError();
}

hipError_t launch_skinned_aabb(const LbsArgs& a, float* d_partials, float* d_out, hipStream_t s) {
    const uint32_t grid = aabb_partial_blocks(a.n_verts);
    hipLaunchKernelGGL(skinned_aabb_kernel, dim3(grid), dim3(kAabbBlock), (size_t)a.n_bones * 64, s, a,
                       d_partials);
    hipLaunchKernelGGL(aabb_final_kernel, dim3(1), dim3(kAabbBlock), 0, s, d_partials, grid, d_out);
    return hipGetLastError();
}

hipError_t launch_points_aabb(const float* d_xyz, uint64_t n_points, float* d_partials, float* d_out,
                              hipStream_t s) {
    const uint32_t grid = aabb_partial_blocks((uint32_t)(n_points > 0xffffffffull ? 0xffffffffu : n_points));
    hipLaunchKernelGGL(points_aabb_kernel, dim3(grid), dim3(kAabbBlock), 0, s, d_xyz, n_points,
                       d_partials);
    hipLaunchKernelGGL(aabb_final_kernel, dim3(1), dim3(kAabbBlock), 0, s, d_partials, grid, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// palette[i] = global[i] * inv_bind[i]   (scene/mesh/mod.rs:781-793; nalgebra gemm order:
// y = a_col0*b_0j ; y = a_colk*b_kj + y).  One thread per output element.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void palette_kernel(const float* __restrict__ A,
                                                      const float* __restrict__ B, uint32_t n,
                                                      float* __restrict__ out) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * 16) return;
    const uint32_t m = e >> 4, j = (e >> 2) & 3, i = e & 3;
    const float* a = A + (size_t)m * 16;
    const float* b = B + (size_t)m * 16;
    float y = a[i] * b[j * 4];
    y = a[4 + i] * b[j * 4 + 1] + y;
    y = a[8 + i] * b[j * 4 + 2] + y;
    y = a[12 + i] * b[j * 4 + 3] + y;
    out[e] = y;
}

hipError_t launch_palette(const float* d_global, const float* d_inv_bind, uint32_t n, float* d_out,
                          hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(palette_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, stream, d_global,
                       d_inv_bind, n, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Control blocks (fyx_ctx.h CtrlBuffers, option anim.ctrl_upload = 2): the frame's few tens of KB read straight out of the pinned
// staging block over the host link and written to the device block -- a kernel launch where a copy command would be.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ctrl_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint32_t n16) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = __builtin_nontemporal_load(src + i);
}

hipError_t launch_ctrl_copy(const void* h_src, void* d_dst, size_t bytes, hipStream_t stream) {
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (!n16) return hipSuccess;
    uint32_t grid = (n16 + 255u) / 256u;
    if (grid > 64u) grid = 64u;     // a few waves' worth of 16-byte reads in flight saturate the link
    if (g_launch_events.start) {     // option debug.timeline
        hipExtLaunchKernelGGL(ctrl_copy_kernel, dim3(grid), dim3(256), 0, stream, g_launch_events.start, g_launch_events.stop, 0,
                              static_cast<const u32x4*>(h_src), static_cast<u32x4*>(d_dst), n16);
        g_launch_events = LaunchEvents();
    } else {
        hipLaunchKernelGGL(ctrl_copy_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const u32x4*>(h_src), static_cast<u32x4*>(d_dst), n16);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Calibration stream: reads 3 float4 planes and writes 2 float4 planes per unit (60 % read /
// 40 % written, the skinning kernel's mix) with the same non-temporal 16-byte accesses and no
// arithmetic.  Its launch time is the achievable ceiling the skinning kernel is compared to,
// and its known byte count calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ src,
                                                          f32x4* __restrict__ dst, uint32_t units) {
    for (uint32_t u = blockIdx.x * 256 + threadIdx.x; u < units; u += gridDim.x * 256) {
        const f32x4 a = __builtin_nontemporal_load(src + u);
        const f32x4 b = __builtin_nontemporal_load(src + (size_t)units + u);
        const f32x4 c = __builtin_nontemporal_load(src + 2 * (size_t)units + u);
        __builtin_nontemporal_store(a + c, dst + u);
        __builtin_nontemporal_store(b - c, dst + (size_t)units + u);
    }
}

hipError_t launch_stream_copy(const float* d_src, float* d_dst, uint32_t units, int blocks_per_cu,
                              hipStream_t stream) {
    if (units == 0) return hipSuccess;
    uint32_t grid = (units + 255) / 256;
    const uint32_t cap = (uint32_t)kCUs * (uint32_t)(blocks_per_cu > 0 ? blocks_per_cu : 8);
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4*>(d_src), reinterpret_cast<f32x4*>(d_dst), units);
    return hipGetLastError();
}

}  // namespace fyx
