/*
 * apm_sieve.hip -- SIEVE + VERIFY: the pipeline of the per-position key classes of the BANDED path
 * (exact for the predicate dist <= k of the reference's window DP, /root/reference/src/utils.c:76-99 applied at
 * every text position by /root/reference/src/sequential.c:105-144; the lemmas are stated in apm_kernels.hip).
 *
 *   apm_sieve2_kernel   one HBM pass over the text, wave-autonomous (no LDS text tile, no barrier in the loop):
 *                       1 KiB chunks, 16 bytes per lane (+ the 8 that follow), four chunks in flight per wave.
 *                       The lane's 24 bytes become a 48-bit string of 2-bit codes (v_dot4_u32_u8 packs four
 *                       codes per instruction); every EVEN position's 18-bit code word (9 bytes) is one lookup in
 *                       a 32 KiB LDS presence bitmap that answers for the position and the odd one behind it.
 *                       The hit masks of a 4 KiB block (32 bits per lane) leave with one coalesced store.
 *   apm_verify_kernel   mask-driven: a wave walks its run of blocks, compacts the hits into batches of 64 (one
 *                       candidate per lane, dense across block borders: the text comes from global memory).  Key identification by rank
 *                       over the exact 16-bit presence bitmap (two dependent LDS reads, no hashing, no tags),
 *                       piece compare + pair pre-check against global text (bounds-checked buffer loads), the
 *                       survivors of a wave are collected and the banded DP + stateless dedup run on dense lanes.
 *
 * Both need a 16-byte aligned text pointer and a shard of < 4 GiB: the runtime scans bigger shards in pieces and falls
 * back to the LDS-tile kernels of apm_kernels.hip for unaligned pointers.
 */
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>
#include "apm_device.h"
#include "apm_sieve.h"

/* tuning constants (each measured on MI355X with tools/ab_libs.sh, one box per comparison) */
#ifndef APM_WORK_CH
#define APM_WORK_CH 8u /* blocks per chunk of the dynamic distribution, per-position sets (2, 8: the same within 3 %; 8 = fewer atomics when most rows are empty) */
#endif
#ifndef APM_WORK_CH8
#define APM_WORK_CH8 8u /* ... sampled sets (8, 16: the same) */
#endif
#ifndef APM_FUSED_NBLK
#define APM_FUSED_NBLK 2u /* fused sampled form without prefetch: blocks per sieve step (2 beats 1 by 4 %, 4 spills) */
#endif
#ifndef APM_FUSED_PREFETCH
#define APM_FUSED_PREFETCH 1 /* fused sampled form: one block per step, the NEXT block's 4 KiB in flight while this one is sieved
                                and its hits verified -- the access shape of the plain sieve kernels (tools/stream_probe.hip: 4 KiB
                                per wave and round streams at 6.2 TB/s, 8 KiB at 4.4) */
#endif
#ifndef APM_VERIFY_PIPE
#define APM_VERIFY_PIPE 2 /* batches formed ahead of the one in hand (2 beats 1 by 17 % on cfg3: the window loads of batch b+1 then do not wait for the queue reads that form it) */
#endif
#ifndef APM_FUSED_PIPE
#define APM_FUSED_PIPE 1 /* the same for the fused form (1 and 2 equal for sampled sets; 2 costs registers) */
#endif

// Raise a kernel's dynamic-LDS limit to the whole CU ONCE per (kernel, device): hipFuncSetAttribute is a host call of tens of
// microseconds, and in front of every launch it showed as kernel time on small inputs (the stream idles while the host works)
static void apm_ensure_max_lds(const void *fn) {
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &d : done)
        if (d.first == fn && d.second == dev) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.emplace_back(fn, dev);
}

typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t apm_lds_u32; // LDS dword, for constant-base accesses

#define APM_SIEVE2_BLOCK 512

__device__ __forceinline__ uint32_t apm_udot4(uint32_t a, uint32_t b) {
    return __builtin_amdgcn_udot4(a, b, 0u, false); // v_dot4_u32_u8
}

// 2-bit codes of 16 bytes (four dwords): byte z of dword q lands in bits 8 q + 2 z.  The code bits are masked where they
// are (byte >> cs is not formed), one v_dot4_u32_u8 per dword leaves (codes << cs), and the shifts go into the combine:
// 12 instructions instead of 15.
__device__ __forceinline__ uint32_t apm_pack16(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t cs) {
#ifdef APM_OLD_PACK /* (A/B builds) */
    auto p4 = [&](uint32_t v) { return apm_udot4((v >> cs) & 0x03030303u, 0x40100401u); };
    return p4(x) | (p4(y) << 8) | (p4(z) << 16) | (p4(w) << 24);
#else
    const uint32_t mask = 0x03030303u << cs;
    const uint32_t p0 = apm_udot4(x & mask, 0x40100401u), p1 = apm_udot4(y & mask, 0x40100401u);
    const uint32_t p2 = apm_udot4(z & mask, 0x40100401u), p3 = apm_udot4(w & mask, 0x40100401u);
    return (p0 >> cs) | (p1 << (8u - cs)) | (p2 << (16u - cs)) | (p3 << (24u - cs));
#endif
}

// ---------------------------------------------------------------------------
// SIEVE
// ---------------------------------------------------------------------------
// CF: with the code filter (ApmSieve2Args::cf_image) -- workgroups of blockDim.x threads, LDS = bitmap | cf image | wave areas
template <bool CF>
__device__ __forceinline__ void apm_sieve2_body(const ApmSieve2Args &a, uint8_t *smem) {
    const int THREADS = CF ? (int)blockDim.x : APM_SIEVE2_BLOCK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= a.n_main_blocks) { // extra workgroups: truncated tail windows (one pattern each)
        apm_tail_body(a.tail, (int)blockIdx.x - a.n_main_blocks, reinterpret_cast<uint4 *>(smem), tid);
        return;
    }
    apm_stage_image(reinterpret_cast<uint4 *>(smem), a.bitmap, 2048, tid, THREADS);
    if constexpr (CF) apm_stage_image(reinterpret_cast<uint4 *>(smem + 32768), a.cf_image, a.cf_len >> 4, tid, THREADS);
    // candidate list (ApmSieve2Args::clist): entries reserved in the workgroup's region | the first reservation that did not fit
    uint32_t *cl_ctr = reinterpret_cast<uint32_t *>(smem + 32768 + (CF ? a.cf_len + (THREADS / 64) * APM_CF_WAVE_BYTES : 0));
    if (CF && tid == 0) { cl_ctr[0] = 0u; cl_ctr[1] = 0xffffffffu; }
    __syncthreads();
    const int64_t W = (int64_t)a.n_main_blocks * (THREADS / 64);
    const int64_t nch = a.nchunks;

    // 16 bytes per lane and chunk; `tl`: the 8 bytes behind the chunk (one address for the whole wave).  Chunks behind the
    // scanned range are loaded all the same -- text or, beyond avail_pad, zeros without traffic -- because the windows of
    // the last valid chunk run into them; only their own hits are dropped.
    // ONE buffer resource over the whole shard (< 4 GiB - 64 MiB, checked by the launcher: the offsets of the chunks a
    // wave loads ahead, up to 4 W beyond the scanned range, do not wrap) and 32-bit offsets: a resource per chunk cost two
    // 64-bit VALU compares per load (the scalar unit has none), ten per block.
    const __amdgpu_buffer_rsrc_t rs_all =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.text), 0, (int)(uint32_t)a.avail_pad, 0x00020000);
    const uint32_t t0_32 = (uint32_t)a.tile0, lane16 = 16u * (uint32_t)lane;
    auto load_chunk = [&](int64_t cc, u32x4 &r) __attribute__((always_inline)) {
        r = __builtin_amdgcn_raw_buffer_load_b128(rs_all, (int)(t0_32 + (uint32_t)cc * 1024u + lane16), 0, 0);
    };
    auto load_tail = [&](int64_t cc, v2u32 &tl) __attribute__((always_inline)) { // the 8 bytes at the start of chunk cc
        tl = __builtin_amdgcn_raw_buffer_load_b64(rs_all, (int)(t0_32 + (uint32_t)cc * 1024u), 0, 0);
    };
    // CF: the halo of the block that starts at chunk cc -- lanes 0, 1: the 32 bytes behind it, lane 2: the 16 bytes in front
    // of it, lane 3 and up: zeros (one load; the resource spans the whole shard: < 4 GiB, zeros outside, and a position in
    // front of the shard wraps to a huge offset = zeros, as for every verify path)
    auto load_halo = [&](int64_t cc, u32x4 &hl) __attribute__((always_inline)) {
        const uint32_t g = (uint32_t)(a.tile0 + cc * 1024);
        const uint32_t off = lane < 2 ? g + 4096u + 16u * (uint32_t)lane : (lane == 2 ? g - 16u : 0xfffffff0u);
        hl = __builtin_amdgcn_raw_buffer_load_b128(rs_all, (int)off, 0, 0);
    };
    // 4 bytes -> 8 code bits (byte z in bits 2z..): shift + and + one v_dot4_u32_u8 with the byte weights 1, 4, 16, 64
    const uint32_t cs = (uint32_t)a.code_shift;
    auto pack4 = [&](uint32_t w4) __attribute__((always_inline)) { return apm_udot4((w4 >> cs) & 0x03030303u, 0x40100401u); };
    auto pack16 = [&](const u32x4 &v) __attribute__((always_inline)) { return apm_pack16(v.x, v.y, v.z, v.w, cs); };
    // the bitmap leads this kernel's LDS (no static LDS, checked by the tests): LDS address = the masked code bits
    // hit mask of the lane's eight even positions: bit 24 + t = position 2t.  slo: codes of the lane's 16 bytes; nx0: of the
    // 8 bytes behind the chunk.  The codes of the 8 bytes behind the LANE's 16 are the low half of the next lane's string:
    // one DPP move (wave_shl:1; the last lane keeps `old` = nx0) instead of a second load and two more packs.
    auto hit_bits = [&](uint32_t slo, uint32_t nx0, int64_t cc) __attribute__((always_inline)) {
        const uint32_t shi = (uint32_t)__builtin_amdgcn_update_dpp((int)nx0, (int)slo, 0x130, 0xf, 0xf, false);
        uint32_t hits = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // y: the 18-bit code word of position 2t in bits 2..19 -> byte address of its bitmap dword = y & 0x7ffc,
            // bit index = bits 15..19 (a shift by a VGPR uses its low five bits)
            const uint32_t y = t ? __builtin_amdgcn_alignbit(shi, slo, 4u * (uint32_t)t - 2u) : (slo << 2);
            const uint32_t word = *(const apm_lds_u32 *)(uintptr_t)(y & 0x7ffcu);
            hits = __builtin_amdgcn_alignbit(word >> ((y >> 15) & 31u), hits, 1u); // bit 0 of the shifted word enters at the top
        }
#ifdef APM_MEASURE
        if (APM_SKIP(a, 1)) hits = 0;
#endif
        return cc < nch ? hits : 0u;
    };

    // ---- CODE FILTER (see ApmSieve2Args::cf_image): which of the block's lookup hits can be a nomination at all ----
    const uint2 *cf_tbl = reinterpret_cast<const uint2 *>(smem + 32768);
    const uint2 *cf_rrec = reinterpret_cast<const uint2 *>(smem + 32768 + a.cf_o_rrec);
    const uint2 *cf_lrec = reinterpret_cast<const uint2 *>(smem + 32768 + a.cf_o_lrec);
    uint32_t *st = reinterpret_cast<uint32_t *>(smem + 32768 + a.cf_len + wv * APM_CF_WAVE_BYTES); // the block's codes: dword 0 =
                                                                 // the 16 bytes in front, 1..256 the block, 257..258 the 32 behind, 259 zero
    uint32_t *mk = st + 260;                                     // surviving hit masks, one dword per lane
    uint16_t *rq = reinterpret_cast<uint16_t *>(mk + 64);        // ring of hits: even position / 2 inside the block | 2048: the odd position only
    // hm: the lane's hit mask of the block (bit 8 j + t = lookup t of chunk j); sc[j]: the codes of its 16 bytes of chunk j;
    // hc: codes of the halo bytes this lane loaded.  Returns the mask of the hits that pass the filter.
    auto cf_filter = [&](uint32_t hm, const uint32_t (&sc)[4], uint32_t hc) __attribute__((always_inline)) -> uint32_t {
        if (!__builtin_amdgcn_ballot_w64(hm != 0u)) return 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) st[1 + 64 * j + lane] = sc[j];
        if (lane < 4) st[lane < 2 ? 257 + lane : (lane == 2 ? 0 : 259)] = hc;
        mk[lane] = 0u;
        uint32_t qh = 0, qt = 0; // wave-uniform: the ring holds entries [qh, qt)
        // nb <= 64 hits, one per lane.  A hit stands for the even position and the odd one behind it: both 16-bit words
        // are looked up; the lane follows the even one if it is a key word, else the odd one; when both are, the odd
        // one goes back into the ring as an entry of its own (rare).
        // Up to 32 hits (most blocks of a sparse set, and the last batch of any block): lanes 0..31 take the even position of
        // entry L, lanes 32..63 the odd one of entry L - 32 -- nothing goes back into the ring, no straggler batch for a
        // handful of odd positions (the both-parities case is common: a unit that tolerates an indel has its neighbour words
        // set too; it cost cfg3 a second batch per block).
        auto run_batch = [&](uint32_t nb) __attribute__((always_inline)) {
#ifdef APM_CF_NOSPLIT /* (A/B builds) */
            const bool split = false;
#else
            const bool split = nb <= 32u; // (wave-uniform)
#endif
            const uint32_t ei = split ? ((uint32_t)lane & 31u) : (uint32_t)lane;
            const uint32_t ent = rq[(qh + ei) & 127u];
            const bool valid = ei < nb;
            const bool take0 = !split || lane < 32, take1 = !split || lane >= 32;
            const uint32_t se = 2u * (ent & 2047u), d = 1u + (se >> 4), she = 2u * (se & 15u); // even byte position inside the block
            const uint32_t w0 = st[d], w1 = st[d + 1u];
            const uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, she) & 0xffffu, x1 = __builtin_amdgcn_alignbit(w1, w0, she + 2u) & 0xffffu;
            const uint2 t0 = cf_tbl[x0 & 2047u], t1 = cf_tbl[x1 & 2047u];
            const bool p0 = valid && take0 && !(ent & 2048u) && ((t0.x >> (x0 >> 11)) & 1u), p1 = valid && take1 && ((t1.x >> (x1 >> 11)) & 1u);
            const unsigned long long both = __builtin_amdgcn_ballot_w64(p0 && p1);
            if (both) { // the odd position waits for a later batch (the ring has room: at most 63 + 64 entries are ever pending)
                const uint32_t idx = qt + __builtin_amdgcn_mbcnt_hi((uint32_t)(both >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)both, 0u));
                if (p0 && p1) rq[idx & 127u] = (uint16_t)(ent | 2048u);
                qt += (uint32_t)__builtin_popcountll(both);
            }
            bool act = p0 || p1;
            const uint32_t s = se + (p0 ? 0u : 1u); // the position this lane judges
            const uint32_t c0 = __builtin_amdgcn_alignbit(w1, w0, she + (p0 ? 0u : 2u)); // codes of the 16 bytes from s on
            const uint32_t x = p0 ? x0 : x1, word = p0 ? t0.x : t1.x, pre = p0 ? t0.y : t1.y, bit = x >> 11;
            uint2 rec = cf_rrec[act ? pre + (uint32_t)__builtin_popcount(word & ((1u << bit) - 1u)) : 0u];
            uint32_t li = 0; // index of the next record of the word's key list
            bool in_list = false;
            if ((rec.x >> 30) == 3u) { li = rec.x & 0xffffu; in_list = true; rec = cf_lrec[li]; ++li; }
            while (__builtin_amdgcn_ballot_w64(act)) {
                const uint32_t side = rec.x >> 30;
                const uint32_t uu = side == 2u ? s : s + 16u + ((rec.y >> 20) & 0xffu); // 16 + the partner's text position
                const bool vis = uu <= 4128u;
                const uint32_t ua = vis ? uu : 16u;
                uint32_t tw = __builtin_amdgcn_alignbit(st[(ua >> 4) + 1u], st[ua >> 4], 2u * (ua & 15u));
                if (side == 2u) tw = apm_rev_codes(tw);
                const bool ok = apm_cf_pass(rec.x, rec.y, c0, tw, vis);
                if (act && ok) {
                    atomicOr(&mk[(s >> 4) & 63u], 1u << (8u * (s >> 10) + ((s & 15u) >> 1)));
                    act = false;
                } else if (act) {
                    if (!in_list || (rec.y >> 31)) act = false;
                    else { rec = cf_lrec[li]; ++li; }
                }
            }
        };
#ifndef APM_CF_NOSCAN /* (A/B builds) */
        // Where the hits go in the ring: a wave prefix sum over the lanes' hit counts (six DPP adds), then every lane writes
        // its own hits one after the other -- a round is ctz + store, not ballot + mbcnt + popcount (cfg5: five rounds per
        // block).  The ring is empty here and holds 128; a fuller block takes the round-by-round form below.
        {
            const uint32_t cnt = (uint32_t)__builtin_popcount(hm);
            uint32_t inc = cnt; // inclusive prefix sum: within the rows of 16 lanes, then across them
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false); // row_shr:1
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false); // row_shr:2
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false); // row_shr:4
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false); // row_shr:8
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (total <= 128u) { // (wave-uniform)
                uint32_t pos = inc - cnt; // (qt == qh == 0: the ring was drained by the block before)
                while (__builtin_amdgcn_ballot_w64(hm != 0u)) {
                    if (hm != 0u) {
                        const uint32_t t = (uint32_t)__builtin_ctz(hm);
                        hm &= hm - 1u;
                        rq[pos & 127u] = (uint16_t)(512u * (t >> 3) + 8u * (uint32_t)lane + (t & 7u));
                        ++pos;
                    }
                }
                qt = total;
                while (qt - qh >= 64u) { run_batch(64u); qh += 64u; }
            }
        }
#endif
        while (__builtin_amdgcn_ballot_w64(hm != 0u)) {
            const bool has = hm != 0u;
            const uint32_t t = has ? (uint32_t)__builtin_ctz(hm) : 0u;
            hm &= hm - 1u;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(has);
            const uint32_t idx = qt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (has) rq[idx & 127u] = (uint16_t)(512u * (t >> 3) + 8u * (uint32_t)lane + (t & 7u));
            qt += (uint32_t)__builtin_popcountll(mask);
            while (qt - qh >= 64u) { run_batch(64u); qh += 64u; }
        }
        while (qt != qh) { const uint32_t nb = qt - qh < 64u ? qt - qh : 64u; run_batch(nb); qh += nb; }
        return mk[lane];
    };

    // the wave's non-empty blocks (ApmSieve2Args::blist): lane i holds the i-th pending block number
    uint32_t bl_pend = 0, bl_n = 0; // (bl_n wave-uniform)
    auto bl_flush = [&]() __attribute__((always_inline)) {
        uint32_t base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add(a.blist_ctr, bl_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((uint32_t)lane < bl_n) a.blist[base + (uint32_t)lane] = bl_pend;
        bl_n = 0;
    };
    if (CF && a.blist && blockIdx.x == 0 && tid == 0) *a.blist_ctr_next = 0u; // (nobody counts in the other set during this launch)

    int64_t c = ((int64_t)blockIdx.x * (THREADS / 64) + wv) * 4; // four neighbouring chunks per wave
    u32x4 r0, r1, r2, r3, hl;
    v2u32 tl;
    load_chunk(c, r0);
    load_chunk(c + 1, r1);
    load_chunk(c + 2, r2);
    load_chunk(c + 3, r3);
    if constexpr (CF) load_halo(c, hl);
    else load_tail(c + 4, tl);
    for (; c < nch; c += 4 * W) {
        const uint32_t s0 = pack16(r0), s1 = pack16(r1), s2 = pack16(r2), s3 = pack16(r3);
        uint32_t s4;
        if constexpr (CF) s4 = pack16(hl); // (lane 0: the low half = the 8 bytes behind the block)
        else s4 = pack4(tl.x) | (pack4(tl.y) << 8);
        load_chunk(c + 4 * W, r0);
        load_chunk(c + 4 * W + 1, r1);
        load_chunk(c + 4 * W + 2, r2);
        load_chunk(c + 4 * W + 3, r3);
        if constexpr (CF) load_halo(c + 4 * W, hl);
        else load_tail(c + 4 * W + 4, tl);
        // (chunk by chunk: letting the scheduler interleave the 32 lookups costs more registers than the 8 waves per SIMD leave)
        const uint32_t h0 = hit_bits(s0, (uint32_t)__builtin_amdgcn_readfirstlane((int)s1), c);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t h1 = hit_bits(s1, (uint32_t)__builtin_amdgcn_readfirstlane((int)s2), c + 1);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t h2 = hit_bits(s2, (uint32_t)__builtin_amdgcn_readfirstlane((int)s3), c + 2);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t h3 = hit_bits(s3, (uint32_t)__builtin_amdgcn_readfirstlane((int)s4), c + 3);
        // the block's hit masks: one coalesced 256-byte store per wave and 4 KiB (see ApmSieve2Args::masks)
        uint32_t hm = (h0 >> 24) | ((h1 >> 24) << 8) | ((h2 >> 24) << 16) | (h3 & 0xff000000u);
        if constexpr (CF) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t sc[4] = {s0, s1, s2, s3};
            hm = cf_filter(hm, sc, s4);
        }
        if constexpr (CF) {
            if (a.clist) { // the survivors leave as list entries, one round per bit of the fullest lane
                const uint32_t e0 = (uint32_t)((a.tile0 + (c >> 2) * 4096) >> 1) + 8u * (uint32_t)lane;
                uint32_t *region = a.clist + (size_t)blockIdx.x * a.clist_cap;
                for (;;) {
                    const bool has = hm != 0u;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(has);
                    if (!mask) break;
                    const uint32_t n = (uint32_t)__builtin_popcountll(mask);
                    uint32_t base = 0;
                    if (lane == 0) base = __hip_atomic_fetch_add(&cl_ctr[0], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base + n > a.clist_cap) { // region full: the reservations before this one are the region's entries
                        if (lane == 0) __hip_atomic_fetch_min(&cl_ctr[1], base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        break;
                    }
                    const uint32_t t = has ? (uint32_t)__builtin_ctz(hm) : 0u;
                    hm &= hm - 1u;
                    if (has)
                        region[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = e0 + 512u * (t >> 3) + (t & 7u);
                }
            }
            // the mask row (all of it without a list, what did not fit with one) and the block's entry in the block list
            const bool any = __builtin_amdgcn_ballot_w64(hm != 0u) != 0ull;
            if (!a.clist || any) a.masks[(size_t)(c >> 2) * 64 + (size_t)lane] = hm;
            if (a.blist && any) {
                if ((uint32_t)lane == bl_n) bl_pend = (uint32_t)(c >> 2);
                if (++bl_n == 64u) bl_flush();
            }
        } else {
            a.masks[(size_t)(c >> 2) * 64 + (size_t)lane] = hm; // (without the filter nearly every block has hits: no list)
        }
    }
    if constexpr (CF) {
        // what is left pending when the wave's run ends leaves with ONE atomic per WORKGROUP: the waves of a launch end
        // together, and 8192 of them adding to one counter took 0.09 ms -- twice the whole sieve of a 256 MiB text (round 3
        // measurement; the adds serialise at ~90 per microsecond).  The per-wave code strips are free by now: each wave
        // parks its pending numbers in its own, thread t then copies entry t % 64 of wave t / 64.
        if (a.blist) { // (workgroup-uniform)
            if ((uint32_t)lane < bl_n) st[lane] = bl_pend;
            if (lane == 0) st[64] = bl_n;
            __syncthreads();
            uint32_t *wg = reinterpret_cast<uint32_t *>(smem + 32768 + a.cf_len); // wave w's strip: wg + w * (APM_CF_WAVE_BYTES / 4)
            const uint32_t nw = (uint32_t)(THREADS / 64);
            uint32_t before = 0, total = 0;
            for (uint32_t w = 0; w < nw; ++w) {
                const uint32_t cw = wg[w * (APM_CF_WAVE_BYTES / 4) + 64];
                if (w < (uint32_t)wv) before += cw;
                total += cw;
            }
            uint32_t *gbase = wg + 65; // (wave 0's strip, behind its own entries)
            if (tid == 0 && total) *gbase = __hip_atomic_fetch_add(a.blist_ctr, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if ((uint32_t)lane < bl_n) a.blist[*gbase + before + (uint32_t)lane] = bl_pend;
            if (a.clist && tid == 0) a.clist_cnt[blockIdx.x] = cl_ctr[0] < cl_ctr[1] ? cl_ctr[0] : cl_ctr[1]; // (behind the barrier: every wave has made its reservations)
        }
    }
}

__global__ __launch_bounds__(APM_SIEVE2_BLOCK, 8) void apm_sieve2_kernel(ApmSieve2Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    apm_sieve2_body<false>(a, smem);
}

// the code-filter form: workgroups of up to 1024 threads share the tables (2 x 16 waves fill a CU)
__global__ __launch_bounds__(1024, 8) void apm_sieve2cf_kernel(ApmSieve2Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    apm_sieve2_body<true>(a, smem);
}

// Sampled form (stride 8): every key piece is >= 15 bytes long and therefore contains an 8-byte block that starts at a
// multiple of 8 in the text; one lookup per 8 text bytes in an 8 KiB bitmap over the blocks' 16-bit code words.  No
// bytes beyond the lane's 16 are needed.  ~25 VALU instructions per KiB: the pass is bound by HBM alone.
__global__ __launch_bounds__(APM_SIEVE2_BLOCK, 8) void apm_sieve8_kernel(ApmSieve2Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= a.n_main_blocks) { // extra workgroups: truncated tail windows (one pattern each)
        apm_tail_body(a.tail, (int)blockIdx.x - a.n_main_blocks, reinterpret_cast<uint4 *>(smem), tid);
        return;
    }
    apm_stage_image(reinterpret_cast<uint4 *>(smem), a.bitmap, 512, tid, APM_SIEVE2_BLOCK);
    __syncthreads();
    const int64_t W = (int64_t)a.n_main_blocks * (APM_SIEVE2_BLOCK / 64);
    const int64_t nch = a.nchunks;
    // (one resource over the shard and 32-bit offsets, as in apm_sieve2_body; chunks behind the range: no load at all)
    const __amdgpu_buffer_rsrc_t rs_all =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.text), 0, (int)(uint32_t)a.avail_pad, 0x00020000);
    const uint32_t t0_32 = (uint32_t)a.tile0, lane16 = 16u * (uint32_t)lane;
    auto load_chunk = [&](int64_t cc, u32x4 &r) __attribute__((always_inline)) {
        r = __builtin_amdgcn_raw_buffer_load_b128(rs_all, (int)(cc < nch ? t0_32 + (uint32_t)cc * 1024u + lane16 : 0xfffffff0u), 0, 0);
    };
    const uint32_t cs = (uint32_t)a.code_shift;
    auto hit_bits = [&](const u32x4 &v, int64_t cc) __attribute__((always_inline)) { // bit t = block at byte 8 t of the lane
        const uint32_t slo = apm_pack16(v.x, v.y, v.z, v.w, cs);
        const uint32_t w0 = *(const apm_lds_u32 *)(uintptr_t)((slo << 2) & 0x1ffcu), w1 = *(const apm_lds_u32 *)(uintptr_t)((slo >> 14) & 0x1ffcu);
        uint32_t hits = ((w0 >> ((slo >> 11) & 31u)) & 1u) | (((w1 >> (slo >> 27)) & 1u) << 1);
#ifdef APM_MEASURE
        if (APM_SKIP(a, 1)) hits = 0;
#endif
        return cc < nch ? hits : 0u;
    };
    int64_t c = ((int64_t)blockIdx.x * (APM_SIEVE2_BLOCK / 64) + wv) * 4; // four neighbouring chunks per wave
    u32x4 r0, r1, r2, r3;
    load_chunk(c, r0);
    load_chunk(c + 1, r1);
    load_chunk(c + 2, r2);
    load_chunk(c + 3, r3);
    for (; c < nch; c += 4 * W) {
        uint32_t h0, h1, h2, h3;
        { const u32x4 v = r0; load_chunk(c + 4 * W, r0); h0 = hit_bits(v, c); }
        { const u32x4 v = r1; load_chunk(c + 4 * W + 1, r1); h1 = hit_bits(v, c + 1); }
        { const u32x4 v = r2; load_chunk(c + 4 * W + 2, r2); h2 = hit_bits(v, c + 2); }
        { const u32x4 v = r3; load_chunk(c + 4 * W + 3, r3); h3 = hit_bits(v, c + 3); }
        // the block's hit masks (two lookups per lane and chunk): one coalesced 256-byte store per wave and 4 KiB
        a.masks[(size_t)(c >> 2) * 64 + (size_t)lane] = h0 | (h1 << 8) | (h2 << 16) | (h3 << 24);
    }
}

static size_t apm_sieve2cf_lds_bytes(int cf_len, int threads) {
    return (size_t)32768 + (size_t)cf_len + (size_t)(threads / 64) * APM_CF_WAVE_BYTES + 16; // (... | the candidate list's two counters)
}

// workgroup size (a multiple of 64) and workgroups per CU that put the most waves on a CU for this code-filter image
int apm_sieve2cf_geometry(int cf_len, int *threads) {
    const void *fn = (const void *)apm_sieve2cf_kernel;
    int best_waves = 0, best_blocks = 0;
    *threads = 0;
    apm_ensure_max_lds(fn);
#ifdef APM_MEASURE
    static const int forced = getenv("APM_CF_THREADS") ? atoi(getenv("APM_CF_THREADS")) : 0;
#else
    constexpr int forced = 0;
#endif
    for (int t = 1024; t >= 256; t -= 64) {
        if (forced && t != forced) continue;
        const size_t lds = apm_sieve2cf_lds_bytes(cf_len, t);
        if (lds > (size_t)160 * 1024) continue;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, t, lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            continue;
        }
        if (per_cu * (t / 64) > best_waves) { // (ties: the bigger workgroup, fewer copies of the tables)
            best_waves = per_cu * (t / 64);
            best_blocks = per_cu;
            *threads = t;
        }
    }
    return best_blocks;
}

int apm_sieve2cf_blocks(const ApmSieve2Args &a, int n_cu) {
    const int wpb = a.cf_threads / 64;
    if (a.nchunks <= 0 || wpb < 1 || a.cf_blocks_per_cu < 1) return 0;
    const int64_t want = (a.nchunks + 4 * wpb - 1) / (4 * wpb), cap = (int64_t)n_cu * a.cf_blocks_per_cu;
    return (int)(want < cap ? want : cap);
}

hipError_t apm_launch_sieve2(const ApmSieve2Args &a, int n_cu, hipStream_t s) {
    if (a.nchunks <= 0) return hipSuccess;
    if (a.avail_pad > APM_SIEVE_MAX_BYTES || a.tile0 < 0) return hipErrorInvalidValue; // (32-bit offsets, the loads a wave issues ahead included)
    ApmSieve2Args args = a;
#ifdef APM_MEASURE
    if (const char *e = getenv("APM_MEASURE_SKIP")) args.skip_mask = atoi(e);
#endif
    void *kargs[] = {&args};
    if (a.stride != 8 && a.cf_image) { // the code-filter form
        const int threads = a.cf_threads;
        if (threads < 128 || threads > 1024 || (threads & 63) || a.cf_blocks_per_cu < 1) return hipErrorInvalidValue;
        const size_t lds = apm_sieve2cf_lds_bytes(a.cf_len, threads);
        const int64_t nb = apm_sieve2cf_blocks(a, n_cu);
        if (a.clist && (!a.blist || !a.clist_cnt || a.clist_cap < 1u)) return hipErrorInvalidValue; // (what does not fit a region leaves through the block list)
        args.n_main_blocks = (int)nb;
        if (lds > 48 * 1024) apm_ensure_max_lds((const void *)apm_sieve2cf_kernel); // (per device: the geometry query ran on one)
        return hipLaunchKernel((const void *)apm_sieve2cf_kernel, dim3((unsigned)(nb + a.n_tail)), dim3((unsigned)threads), kargs, lds, s);
    }
    const size_t lds = 32768;
    const int64_t want = (a.nchunks + 4 * (APM_SIEVE2_BLOCK / 64) - 1) / (4 * (APM_SIEVE2_BLOCK / 64));
    const int64_t cap = (int64_t)n_cu * 4; // = the kernel's launch bound (4 x 512 threads per CU; 4 x 32 KiB of LDS)
    const int64_t nb = want < cap ? want : cap;
    args.n_main_blocks = (int)nb;
    if (a.stride == 8)
        return hipLaunchKernel((const void *)apm_sieve8_kernel, dim3((unsigned)(nb + a.n_tail)), dim3(APM_SIEVE2_BLOCK), kargs, 8192, s);
    return hipLaunchKernel((const void *)apm_sieve2_kernel, dim3((unsigned)(nb + a.n_tail)), dim3(APM_SIEVE2_BLOCK), kargs, lds, s);
}

// ---------------------------------------------------------------------------
// VERIFY
// ---------------------------------------------------------------------------
// text of the shard behind a bounds-checked buffer resource: bytes at or beyond avail_pad (and "negative"
// positions, which wrap to huge offsets) read as zero, for every path alike
struct ApmBufText {
    __amdgpu_buffer_rsrc_t rs;
    uint32_t off; // window start (relative position)
    static constexpr bool kBlocks = true; // the DP fetches its columns 16 at a time (apm_banded_verify)
    __device__ __forceinline__ bool can16(int) const { return true; }
    __device__ __forceinline__ void load16(uint32_t (&T)[4]) const { load16_at(0, T); }
    __device__ __forceinline__ void load16_at(int x0, uint32_t (&T)[4]) const {
        const uint32_t a0 = (off + (uint32_t)x0) & ~3u, sh = (off + (uint32_t)x0) & 3u;
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a0, 0, 0);
        const uint32_t hi = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(a0 + 16u), 0, 0);
        T[0] = __builtin_amdgcn_alignbyte(lo.y, lo.x, sh);
        T[1] = __builtin_amdgcn_alignbyte(lo.z, lo.y, sh);
        T[2] = __builtin_amdgcn_alignbyte(lo.w, lo.z, sh);
        T[3] = __builtin_amdgcn_alignbyte(hi, lo.w, sh);
    }
    __device__ __forceinline__ int byte(int x) const { return (int)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)(off + (uint32_t)x), 0, 0); }
};

// six dwords of text from a 4-byte aligned position a0: bytes [a0, a0 + 24)
struct ApmWin { uint32_t w[6]; };

// The verification core shared by the list-driven verify kernel and the fused kernel: the nomination predicate of a
// unit, the banded DP of the window it implies, and the stateless dedup of matches.  Text comes through a bounds-checked
// buffer resource (zeros outside the shard); the predicate takes the loader of its partner's text as a parameter (the
// fused kernel reads it out of its LDS copy of the block).
template <int BAND>
struct ApmVerifyCore {
    static constexpr int NSH = 2 * BAND + 1;
    static constexpr bool PAIRS = BAND >= 1;
    const ApmVerifyArgs &a;
    __amdgpu_buffer_rsrc_t rs;
    uint32_t avail;
    const uint32_t *s_kext;
    const uint4 *s_masks;
    const uint8_t *s_pat;
    uint32_t *s_cnt;
    int lane;
    const uint32_t *s_kinfo; // per key (LDS: the DP and the dedup are a chain of dependent reads, and with the sieve's code
    const uint2 *s_pinfo;    // filter in front they are most of what the launch does); per pattern

    __device__ __forceinline__ void load_global(uint32_t a0, ApmWin &o) const {
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a0, 0, 0);
        const v2u32 hi = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(a0 + 16u), 0, 0);
        o.w[0] = lo.x; o.w[1] = lo.y; o.w[2] = lo.z; o.w[3] = lo.w; o.w[4] = hi.x; o.w[5] = hi.y;
    }
    __device__ __forceinline__ int gbyte(uint32_t pos) const { // (slow paths only)
        return (int)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)pos, 0, 0);
    }

    // ---- the nomination predicate: key `kid` (one pigeonhole piece) at text position s --------------------
    // piece intact at s, entirely inside the valid text, and (k >= 2) its partner of the pair pre-check within one
    // edit (see apm_kernels.hip, "hierarchical verification").  `win` = the six text dwords at s & ~3.
    // ONE definition for the candidates of the list and for the dedup's "earlier nominator" test.
    template <typename LoadWin>
    __device__ __forceinline__ bool stage1(uint32_t kid, uint32_t s, const ApmWin &win, LoadWin &&load_win) const {
        typedef unsigned long long u64;
        const uint32_t kx = s_kext[kid];
        const int at = (int)(kx & 0xffffu), len = (int)((kx >> 16) & 0xffu), n = (int)((kx >> 24) & 31u), side = (int)(kx >> 29);
        if ((u64)s + (u64)len > (u64)avail) return false;
        const uint32_t sh = s & 3u;
        uint32_t A[4], B[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) A[i] = __builtin_amdgcn_alignbyte(win.w[i + 1], win.w[i], sh); // text bytes [s, s+16)
        apm_lds_dwords<4>(s_pat, at, B);
        const uint4 mk = s_masks[len < 16 ? len : 16]; // 0xff for the first min(len, 16) bytes
        if ((((A[0] ^ B[0]) & mk.x) | ((A[1] ^ B[1]) & mk.y) | ((A[2] ^ B[2]) & mk.z) | ((A[3] ^ B[3]) & mk.w)) != 0u) return false; // the exact part is not intact
        for (int x = 16; x < len; ++x)       // (pieces beyond 16 bytes: patterns with long pieces in this class)
            if (gbyte(s + (uint32_t)x) != (int)s_pat[at + x]) return false;
        if (!PAIRS || side == 0) return true; // no pre-check (k <= 1) / unpaired last piece (even k)
        if (n == 31) { // partner longer than 16 bytes: byte loops (definition of the core, apm_ext_fwd / apm_ext_bwd)
            const uint32_t kp = a.kpart[kid];
            const int poff = (int)(s_pinfo[s_kinfo[kid] & 0xfffu].x & 0xffffu), ap = (int)(kp & 0xffffu), nn = (int)(kp >> 16), ap1 = ap + nn;
            const bool fwd = side == 1;
            auto T = [&](int i) { return fwd ? gbyte(s + (uint32_t)len + (uint32_t)i) : gbyte(s - 1u - (uint32_t)i); };      // text, read away from the exact part
            auto P = [&](int i) { return fwd ? (int)s_pat[poff + ap + i] : (int)s_pat[poff + ap1 - 1 - i]; };                // partner, same direction
            int i = 0;
            while (i < nn && T(i) == P(i)) ++i;
            if (i >= nn - 1) return true;
            bool ok = true;
            for (int j = i + 1; j < nn && ok; ++j) ok = T(j) == P(j);
            if (ok) return true;
            ok = true;
            for (int j = i + 1; j < nn && ok; ++j) ok = T(j - 1) == P(j);
            if (ok) return true;
            ok = true;
            for (int j = i; j < nn && ok; ++j) ok = T(j + 1) == P(j);
            return ok;
        }
        uint32_t P[4], T[5];
        ApmWin tw;
        if (side == 1) { // partner behind the piece: text read forward from the end of the piece
            const uint32_t tp = s + (uint32_t)len;
            if (len == 0 || APM_SKIP(a, 1024)) tw = win; // (a pair of short pieces as one unit: its text starts at s itself)
            else load_win(tp & ~3u, tw);
            apm_lds_dwords<4>(s_pat, at + len, P);
#pragma unroll
            for (int i = 0; i < 5; ++i) T[i] = __builtin_amdgcn_alignbyte(tw.w[i + 1], tw.w[i], tp & 3u);
        } else { // partner in front of it: both strings byte-reversed, text = the 20 bytes in front of s
            uint32_t Q[4], Wd[5];
            apm_lds_dwords<4>(s_pat, at - 16, Q);
            if (s >= 20u) {
                const uint32_t tp = s - 20u;
                if (APM_SKIP(a, 1024)) tw = win; // (measurement: what the dependent gather costs)
                else load_win(tp & ~3u, tw);
#pragma unroll
                for (int i = 0; i < 5; ++i) Wd[i] = __builtin_amdgcn_alignbyte(tw.w[i + 1], tw.w[i], tp & 3u);
            } else { // the first 20 positions of the shard: bytes in front of text[0] do not exist and read as zero
#pragma unroll
                for (int i = 0; i < 5; ++i) Wd[i] = 0u;
                for (int i = 20 - (int)s; i < 20; ++i) {
                    const uint32_t b = (uint32_t)gbyte(s - 20u + (uint32_t)i);
#pragma unroll
                    for (int d = 0; d < 5; ++d)
                        if ((i >> 2) == d) Wd[d] |= b << (8 * (i & 3));
                }
            }
#pragma unroll
            for (int z = 0; z < 4; ++z) P[z] = apm_bswap(Q[3 - z]);
#pragma unroll
            for (int z = 0; z < 5; ++z) T[z] = apm_bswap(Wd[4 - z]);
        }
        // necessary first: the partner's first four bytes within one edit (a prefix of an alignment with <= 1 edit has
        // <= 1 edit): nonzero-byte masks of P ^ T under the three alignments, 4 bits each; rejects ~9 of 10 random texts
        if (n >= 4) {
            auto nz4 = [](uint32_t x) { return apm_udot4((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7 & 0x01010101u, 0x08040201u); };
            const uint32_t z0 = nz4(P[0] ^ T[0]);
            if (z0 & (z0 - 1u)) { // two or more mismatching bytes under the substitution alignment
                const uint32_t i = (uint32_t)__builtin_ctz(z0); // first mismatching byte: 0..2
                const uint32_t z1 = nz4(P[0] ^ (T[0] << 8));                                  // pattern byte i has no text counterpart
                const uint32_t z2 = nz4(P[0] ^ __builtin_amdgcn_alignbyte(T[1], T[0], 1u));   // one extra text byte before pattern byte i
                if (((z1 >> (i + 1u)) != 0u) && ((z2 >> i) != 0u)) return false;
            }
        }
        return apm_ext1_core16(P, T, n);
    }


    // ---- banded DP of the window a nomination (unit kid at text position s) implies under shift dl ----
    // on a match: wpat = pattern slot, wj = window start, word = rank of (unit, shift) among the window's nominators
    __device__ __forceinline__ bool dp_match(uint32_t kid, uint32_t s, int dl, uint32_t &wpat, uint32_t &wj, uint32_t &word) const {
        const uint32_t ki = s_kinfo[kid];
        const int kpat = (int)(ki & 0xfffu), koff = (int)((ki >> 12) & 0x1ffu), kunit = (int)((ki >> 21) & 7u);
        const uint2 pinf = s_pinfo[kpat];
        const int poff = (int)(pinf.x & 0xffffu), m = (int)(pinf.x >> 16);
        const int64_t je_p = min(a.je, a.nrel - m + 1);
        const int64_t j = (int64_t)s - koff - dl; // candidate window start
        if (j < a.jb || j >= je_p) return false;
        wpat = (uint32_t)kpat;
        wj = (uint32_t)j;
        word = (uint32_t)(kunit * NSH + dl + BAND);
#ifdef APM_MEASURE
        if (APM_SKIP(a, 256)) atomicAdd(&a.stats[2], 1ull); // (bit 8 = collect the statistics: one atomic per DP item distorts the timing)
#endif
        return apm_banded_verify<BAND>(ApmBufText{rs, (uint32_t)j}, s_pat, poff, m, a.k);
    }

    // ---- stateless dedup: a matching window counts only from its FIRST true (unit, shift) nominator.  Matches are
    // rare but come in bursts (an occurrence is nominated by every intact unit, its neighbour windows match too, and
    // they all sit in one wave).  Among the matches of a round a window is kept by its smallest (unit, shift) only;
    // what is left is resolved by the whole wave, one match at a time, one lane per earlier (unit, shift): up to
    // 8 x NSH predicate evaluations with their own text fetches, side by side. ----
    __device__ __forceinline__ void count_matches(bool hit, uint32_t wpat, uint32_t wj, uint32_t word) const {
        for (unsigned long long m2 = __builtin_amdgcn_ballot_w64(hit); m2; m2 &= m2 - 1ull) {
            const int src = __builtin_ctzll(m2);
            const uint32_t bpat = (uint32_t)__builtin_amdgcn_readlane((int)wpat, src), bj = (uint32_t)__builtin_amdgcn_readlane((int)wj, src);
            const uint32_t bord = (uint32_t)__builtin_amdgcn_readlane((int)word, src);
            if (hit && wpat == bpat && wj == bj && word > bord) hit = false;
        }
        unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
        while (hm) {
            const int src = __builtin_ctzll(hm);
            hm &= hm - 1ull;
            const uint32_t bpat = (uint32_t)__builtin_amdgcn_readlane((int)wpat, src), bj = (uint32_t)__builtin_amdgcn_readlane((int)wj, src);
            const int n_before = __builtin_amdgcn_readlane((int)word, src); // (unit, shift) pairs in front of this one: < 64
            const uint32_t kid0 = s_pinfo[bpat].y; // the pattern's first unit
            bool earlier = false;
#ifdef APM_MEASURE
            if (APM_SKIP(a, 32)) continue;
#endif
            if (lane < n_before) {
                const int qq = lane / NSH, dd = lane % NSH - BAND;
                const int64_t o = (int64_t)bj + (int)((s_kinfo[kid0 + (uint32_t)qq] >> 12) & 0x1ffu) + dd; // the unit's text position under shift dd
                if (o >= 0) {
                    ApmWin w2;
                    load_global((uint32_t)o & ~3u, w2);
                    earlier = stage1(kid0 + (uint32_t)qq, (uint32_t)o, w2, [&](uint32_t a0, ApmWin &o2) { load_global(a0, o2); });
                }
            }
            if (!__builtin_amdgcn_ballot_w64(earlier) && lane == 0) {
                atomicAdd(&s_cnt[bpat], 1u);
#ifdef APM_MEASURE
                if (APM_SKIP(a, 256)) atomicAdd(&a.stats[3], 1ull);
#endif
            }
        }
    }

};

__host__ __device__ constexpr int apm_verify_scap(int band) { return ((64 + 2 * band) / (2 * band + 1) + 63 + 7) & ~7; }

// THREADS = 256 or 512: the bigger workgroup shares one LDS image among eight waves -- more waves per CU when the
// image (many keys) limits the workgroups per CU
// SAMPLED: the list comes from the stride-8 sieve (see ApmVerifyArgs::stride)
// FUSED: the hit masks do not come from a sieve launch -- the wave sieves its blocks itself (sv: the sieve's arguments;
// stride 1: its 32 KiB bitmap leads the LDS, the image follows; stride 8: the image's own bitmap is the sieve's) and
// verifies the hits in the same batches of 64 across block borders, the windows gathered from global memory (L2 /
// Infinity Cache: the wave streamed those lines a moment ago).  One launch, the text leaves HBM once, no masks.
// THREADS_T = 0: the workgroup size is the launch's (a multiple of 64).
template <int BAND, int THREADS_T, bool SAMPLED, bool FUSED>
__device__ __forceinline__ void apm_verify_body(const ApmVerifyArgs &a, const ApmSieve2Args *sv, uint8_t *smem) {
    const int THREADS = THREADS_T ? THREADS_T : (int)blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NSH = 2 * BAND + 1;
    constexpr uint32_t FLUSH_AT = (64 + NSH - 1) / NSH; // survivors that fill a wave of (survivor, shift) items (capacity of the list: apm_verify_scap)
    (void)FLUSH_AT;
    constexpr int SCAP = apm_verify_scap(BAND);         // capacity of a wave's survivor list: FLUSH_AT - 1 + one round of 64
    uint8_t *s_img = smem + ((FUSED && !SAMPLED) ? 32768 : 0);
    const uint32_t *s_bmp = reinterpret_cast<const uint32_t *>(s_img);
    const uint16_t *s_prefix = reinterpret_cast<const uint16_t *>(s_img + a.o_prefix);
    const uint16_t *s_r2s = reinterpret_cast<const uint16_t *>(s_img + a.o_r2s);
    const uint16_t *s_slots = reinterpret_cast<const uint16_t *>(s_img + a.o_slots);
    const uint32_t *s_kext = reinterpret_cast<const uint32_t *>(s_img + a.o_kext);
    const uint8_t *s_pat = s_img + a.o_pat;
    const uint4 *s_masks = reinterpret_cast<const uint4 *>(s_img + a.o_masks);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_img + a.image_len);
    uint2 *s_surv = reinterpret_cast<uint2 *>(s_cnt + ((a.n_pats + 3) & ~3)) + wv * SCAP; // this wave's survivors {position, kid}
    uint32_t *s_q = reinterpret_cast<uint32_t *>(reinterpret_cast<uint2 *>(s_cnt + ((a.n_pats + 3) & ~3)) + (THREADS / 64) * SCAP) + wv * 128; // this wave's hit queue

    if constexpr (FUSED && !SAMPLED)
        apm_stage_image(reinterpret_cast<uint4 *>(smem), sv->bitmap, 2048, tid, THREADS);
    apm_stage_image(reinterpret_cast<uint4 *>(s_img), a.image, a.image_len >> 4, tid, THREADS);
    for (int i = tid; i < a.n_pats; i += THREADS) s_cnt[i] = 0u;
    __syncthreads(); // the only workgroup barrier before the final count flush

    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.text), 0, (int)(uint32_t)a.avail_pad, 0x00020000);
    const uint32_t avail = (uint32_t)a.avail;
    const uint32_t cs = (uint32_t)a.code_shift;

    ApmVerifyCore<BAND> core{a, rs, avail, s_kext, s_masks, s_pat, s_cnt, lane, reinterpret_cast<const uint32_t *>(s_img + a.o_kinfo),
                             reinterpret_cast<const uint2 *>(s_img + a.o_pinfo)};
    typedef ApmWin Win;
    auto load_win = [&](uint32_t a0, Win &o) __attribute__((always_inline)) { core.load_global(a0, o); };

    // ---- batches of 64 candidates per wave.  One loop, one stage-1 site, one DP site: each trip either runs the
    // DP pass over the wave's survivor list, or moves to the next (batch, parity), or evaluates the predicate once
    // for every lane that still has a key to try at its position. ----
    uint32_t n_surv = 0; // wave-uniform
    uint32_t item_lo = 0; // wave-uniform: shifts of the list's first survivor that an earlier DP pass has taken already
    // the sieve's 4 KiB blocks are dealt to the waves in equal contiguous runs; a wave compacts the hit masks of its
    // blocks (one dword per lane and block) into a queue of positions and takes 64 of them per batch -- dense lanes
    // across block borders, since the text comes from global memory anyway
    constexpr uint32_t STEP = SAMPLED ? 8u : 2u; // bytes between two lookups of the sieve
    const uint32_t my_wave = blockIdx.x * (uint32_t)(THREADS / 64) + (uint32_t)wv;
    // ---- which blocks a wave works on: DYNAMIC.  Equal static runs left the waves finishing anywhere between 0.45 and
    // 1.0 of the kernel's duration (per-wave time stamps, measurement build).  The blocks form chunks of CH; the chunks
    // are split into APM_WORK_GROUPS contiguous ranges, each with its own counter (a single one would serialise: ~90
    // atomics per microsecond chip-wide); wave w belongs to group w % APM_WORK_GROUPS -- every group is a sample of the
    // whole machine, so the groups finish together -- and takes the group's next chunk with one atomic, issued a
    // chunk ahead of its use.  The counters of the NEXT launch are zeroed here (two sets, the host alternates). ----
    // 4 KiB blocks in all; with a block list (ApmVerifyArgs::blist) only the listed ones: entry b of the list is the block
    // -- when the list is short: with most blocks on it (cfg3) the walk over all rows is the shorter chain of loads
    const uint32_t n_listed = (!FUSED && a.blist != nullptr) ? *a.blist_ctr : 0xffffffffu;
    const bool listed = n_listed < (uint32_t)a.n_mask_blocks / 4u || (!FUSED && a.clist != nullptr); // (with a candidate list only the listed blocks have rows at all)
    const uint32_t NB = FUSED ? (uint32_t)((sv->nchunks + 3) >> 2) : (listed ? n_listed : (uint32_t)a.n_mask_blocks);
    // blocks per chunk: a short list is dealt block by block (cfg5: 10 K listed blocks for 4 K waves -- with chunks of 8 most
    // waves got none and the rest walked theirs one load after the other: 0.072 ms against 0.036)
    // (and a short text in chunks small enough that every wave gets a few: 64 MiB in chunks of 8 left two waves of three idle)
    const uint32_t n_waves_launch = (uint32_t)(FUSED ? sv->n_main_blocks : (int)gridDim.x) * (uint32_t)(THREADS / 64);
    const uint32_t ch_fit = NB / (4u * n_waves_launch);
    const uint32_t CH = SAMPLED ? APM_WORK_CH8 : (listed ? 1u : (ch_fit >= APM_WORK_CH ? APM_WORK_CH : (ch_fit < 1u ? 1u : ch_fit)));
    const uint32_t NC = (NB + CH - 1u) / CH;
    // NG = min(APM_WORK_GROUPS, waves of the launch): no group without a wave.  Workgroups go round the XCDs, so the low
    // bits of the wave number alone would tie a group to one XCD and one wave slot: fold the higher bits in
    const uint32_t NG = (uint32_t)a.work_groups, grp = (my_wave ^ (my_wave >> 5)) % NG;
    const uint32_t f_g = (uint32_t)(((uint64_t)NC * grp) / NG), n_g = (uint32_t)(((uint64_t)NC * (grp + 1u)) / NG) - f_g;
    uint32_t *const ctr = a.work + ((uint32_t)a.work_epoch & 1u) * (APM_WORK_GROUPS * APM_WORK_STRIDE) + grp * APM_WORK_STRIDE;
    if (blockIdx.x == 0 && tid < APM_WORK_GROUPS) a.work[(((uint32_t)a.work_epoch + 1u) & 1u) * (APM_WORK_GROUPS * APM_WORK_STRIDE) + (uint32_t)tid * APM_WORK_STRIDE] = 0u;
    auto grab = [&]() __attribute__((always_inline)) -> uint32_t { // (lane 0 holds the answer; read with readfirstlane when it is needed)
        return lane == 0 ? __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    };
    uint32_t grab_v = grab();
    uint32_t it_b = 0, it_end = 0; // the chunk in hand: blocks [it_b, it_end)
    bool it_done = false;
#ifndef APM_FUSED_STATIC
#define APM_FUSED_STATIC 1 /* fused sampled form: blocks dealt statically, wave w takes blocks w, w + W, ... -- with the register
                              compare in front next to nothing is left to verify, so there is nothing to balance, and the
                              waves of a round read one contiguous stretch of text (the plain sieve's access shape) */
#endif
    constexpr bool STATIC_BLOCKS = FUSED && SAMPLED && APM_FUSED_STATIC;
    const uint32_t n_waves_all = (uint32_t)(FUSED ? sv->n_main_blocks : (int)gridDim.x) * (uint32_t)(THREADS / 64); // (the scanning workgroups: not the tail ones)
    uint32_t st_b = my_wave;
    auto it_next = [&](uint32_t &b) __attribute__((always_inline)) -> bool { // wave-uniform: the wave's next block
        if constexpr (STATIC_BLOCKS) {
            if (st_b >= NB) return false;
            b = st_b;
            st_b += n_waves_all;
            return true;
        }
        if (it_b >= it_end) {
            if (it_done) return false;
            const uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)grab_v);
            if (i >= n_g) { it_done = true; return false; } // every wave gets here: its group's range is exhausted
            it_b = (f_g + i) * CH;
            it_end = it_b + CH < NB ? it_b + CH : NB;
            grab_v = grab();
        }
        b = it_b++;
        return true;
    };
    // masks of the block in hand and of the AHEAD blocks after it (sparse sampled lists are bound by this chain of loads)
#ifndef APM_VERIFY_AHEAD
#define APM_VERIFY_AHEAD 2 /* mask rows in flight per wave in front of the block in hand, per-position sets (1, 2, 4: the same within the box-to-box noise; 4 spills in the 72-register instantiation) */
#endif
    constexpr int AHEAD = FUSED ? 1 : (SAMPLED ? 4 : APM_VERIFY_AHEAD);
    constexpr uint32_t NONE = 0xffffffffu;
    uint32_t hm = 0, hm_q[AHEAD], hb_q[AHEAD]; // hb_q: their block numbers (wave-uniform)
    if constexpr (!FUSED) {
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) {
            uint32_t b = NONE;
            hb_q[i] = it_next(b) ? (listed ? a.blist[b] : b) : NONE;
            hm_q[i] = hb_q[i] != NONE ? a.masks[(uint64_t)hb_q[i] * 64 + (uint64_t)lane] : 0u;
        }
    }
#ifdef APM_MEASURE
    if (APM_SKIP(a, 512) && lane == 0 && my_wave < APM_STATS_WAVES) a.stats[8 + 2 * my_wave] = wall_clock64();
#endif
    uint32_t blk = 0;    // relative position of the block in hand
    uint32_t qcount = 0; // wave-uniform
    const uint32_t nch32 = FUSED ? (uint32_t)sv->nchunks : 0u;
    const uint32_t tile0 = FUSED ? (uint32_t)sv->tile0 : 0u;
    auto pack4 = [&](uint32_t w4) __attribute__((always_inline)) { return apm_udot4((w4 >> cs) & 0x03030303u, 0x40100401u); };
    // hit mask of this lane for the block at relative position b0 (see ApmSieve2Args::masks for the bit layout)
    // FUSED + SAMPLED: a sieve step takes NBLK neighbouring blocks (a block fills 8 of the 32 mask bits: block i of the step
    // sits in bits 8 j + 2 i + t) -- twice the bytes in flight per wave; the pass is bound by the latency of these loads
    constexpr bool PREF = FUSED && SAMPLED && APM_FUSED_PREFETCH;
    constexpr uint32_t NBLK = (FUSED && SAMPLED && !PREF) ? APM_FUSED_NBLK : 1u;
    u32x4 pf_r[4];      // PREF: the prefetched block's text
    uint32_t pf_sl[4];  // ... the codes of the block in hand (packed before the next block's loads go out: no second copy of the text)
    uint32_t pf_b = 0xffffffffu; // the prefetched block (none)
    bool pf_started = false;
    auto pf_issue = [&](uint32_t b) __attribute__((always_inline)) {
        const uint32_t g = tile0 + b * 4096u + 16u * (uint32_t)lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) pf_r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(g + 1024u * j), 0, 0); // (beyond the text: zeros)
    };
    auto sieve_block = [&](uint32_t b0, uint32_t fb, uint32_t nblk) __attribute__((always_inline)) -> uint32_t {
        const uint32_t g = b0 + 16u * (uint32_t)lane, c0 = fb * 4u;
        uint32_t out = 0;
        if constexpr (SAMPLED) { // one lookup per 8 bytes in the image's bitmap over 16-bit code words (apm_sieve8_kernel)
            uint32_t sl[4 * NBLK]; // codes of the lane's 16 bytes, chunk by chunk
            if constexpr (PREF) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sl[j] = pf_sl[j];
            } else {
                u32x4 r[4 * NBLK];
#pragma unroll
                for (int j = 0; j < (int)(4 * NBLK); ++j) r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(g + 1024u * j), 0, 0); // (beyond the text: zeros)
#pragma unroll
                for (int j = 0; j < (int)(4 * NBLK); ++j) sl[j] = apm_pack16(r[j].x, r[j].y, r[j].z, r[j].w, cs);
            }
#pragma unroll
            for (int j = 0; j < (int)(4 * NBLK); ++j) {
                const uint32_t slo = sl[j];
                const uint32_t w0 = s_bmp[slo & 2047u], w1 = s_bmp[(slo >> 16) & 2047u];
                const uint32_t h = ((w0 >> ((slo >> 11) & 31u)) & 1u) | (((w1 >> (slo >> 27)) & 1u) << 1);
                out |= ((c0 + j < nch32 && (uint32_t)(j >> 2) < nblk) ? h : 0u) << (8 * (j & 3) + 2 * (j >> 2));
            }
            // REGISTER COMPARE: a hit says "an 8-byte block of some key's piece, r bytes into the piece, may lie here"; the
            // piece is >= 15 bytes long, so more of it lies inside the lane's own 16 bytes -- compare the codes of that overlap
            // (pattern bytes out of the LDS image, packed like the text) before the hit is queued.  What was queued before
            // -- 0.8 M hits per GiB for cfg4, practically all false, each with two window gathers that missed the caches
            // (1.18 x the text in HBM traffic) -- no longer leaves the lane.  A filter (codes equal is necessary for the
            // piece to be intact at position - r, stage1's first test); lanes work on their own hits, one key at a time.
#ifndef APM_NO_REGCMP /* (A/B builds: tools/build_variant.sh) */
            {
                uint32_t pend = out, cur = 0, curbit = 0, slo_c = 0, tsel = 0;
                bool act = false;
                for (;;) {
                    if (!act && pend) { // the lane's next hit: key list of its code word by rank
                        curbit = (uint32_t)__builtin_ctz(pend);
                        pend &= pend - 1u;
                        tsel = curbit & 1u;
                        const uint32_t j = (curbit >> 3) + 4u * ((curbit >> 1) & 3u);
#pragma unroll
                        for (int q = 0; q < (int)(4 * NBLK); ++q)
                            if (j == (uint32_t)q) slo_c = sl[q];
                        const uint32_t x = tsel ? (slo_c >> 16) : (slo_c & 0xffffu), bit = x >> 11;
                        const uint32_t word = s_bmp[x & 2047u];
                        const uint32_t e = s_r2s[(uint32_t)s_prefix[x & 2047u] + (uint32_t)__builtin_popcount(word & ((1u << bit) - 1u))];
                        cur = (e & 0x8000u) ? e : ((uint32_t)s_slots[e] | ((e + 1u) << 16));
                        act = true;
                    }
                    if (!__builtin_amdgcn_ballot_w64(act)) break;
                    if (act && a.o_rc) { // (wave-uniform choice: small sets carry the compare's operands ready made)
                        const uint32_t kid = cur & 2047u, rr = (cur & 0x7fffu) >> 11;
                        const uint2 rc = reinterpret_cast<const uint2 *>(s_img + a.o_rc)[(kid * 8u + rr) * 2u + tsel];
                        if (((rc.x ^ slo_c) & rc.y) == 0u) act = false; // may be intact: the hit stays
                        else if (cur & 0x8000u) { act = false; out &= ~(1u << curbit); } // no key of the word fits
                        else cur = (uint32_t)s_slots[cur >> 16] | ((cur & 0xffff0000u) + 0x10000u);
                    } else if (act) {
                        const uint32_t kid = cur & 2047u, rr = (cur & 0x7fffu) >> 11; // (KBITS = 11: key id | offset of the block in its piece)
                        const uint32_t kx = s_kext[kid];
                        const int at = (int)(kx & 0xffffu), len = (int)((kx >> 16) & 0xffu);
                        // lane byte i <-> piece byte i - 8 t + r <-> pattern pool byte at + r - 8 t + i
                        const int sh8 = (int)(8u * tsel) - (int)rr;
                        uint32_t B[4];
                        apm_lds_dwords<4>(s_pat, at - sh8, B);
                        const uint32_t pc = pack4(B[0]) | (pack4(B[1]) << 8) | (pack4(B[2]) << 16) | (pack4(B[3]) << 24);
                        const int i0 = sh8 > 0 ? sh8 : 0, i1 = len + sh8 < 16 ? len + sh8 : 16; // the piece covers lane bytes [i0, i1)
                        const uint32_t mhi = i1 >= 16 ? 0xffffffffu : ((1u << (2 * i1)) - 1u), mlo = (1u << (2 * i0)) - 1u;
                        if ((((pc ^ slo_c) & mhi) & ~mlo) == 0u) act = false; // may be intact: the hit stays
                        else if (cur & 0x8000u) { act = false; out &= ~(1u << curbit); } // no key of the word fits
                        else cur = (uint32_t)s_slots[cur >> 16] | ((cur & 0xffff0000u) + 0x10000u);
                    }
                }
            }
#endif
        } else { // one lookup per even position in the 32 KiB bitmap over 18-bit code words at LDS address 0 (apm_sieve2_kernel)
            // (chunks behind the scanned range are loaded all the same -- text or zeros -- since the windows of the last
            // valid chunk run into them; only their own hits are dropped)
            u32x4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(g + 1024u * j), 0, 0);
            const v2u32 tl = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(b0 + 4096u), 0, 0); // the 8 bytes behind the block
            uint32_t slo[5]; // codes of the lane's 16 bytes, chunk by chunk; [4]: of the 8 bytes behind the block
#pragma unroll
            for (int j = 0; j < 4; ++j) slo[j] = apm_pack16(r[j].x, r[j].y, r[j].z, r[j].w, cs);
            slo[4] = pack4(tl.x) | (pack4(tl.y) << 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // codes of the 8 bytes behind the lane's 16 = the low half of the next lane's string (lane 63: of the
                // next chunk's lane 0): one DPP move (wave_shl:1; the last lane keeps `old`)
                const uint32_t nx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slo[j + 1]);
                const uint32_t shi = (uint32_t)__builtin_amdgcn_update_dpp((int)nx0, (int)slo[j], 0x130, 0xf, 0xf, false);
                uint32_t hits = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const uint32_t y = t ? __builtin_amdgcn_alignbit(shi, slo[j], 4u * (uint32_t)t - 2u) : (slo[j] << 2);
                    const uint32_t word = *(const apm_lds_u32 *)(uintptr_t)(y & 0x7ffcu);
                    hits = __builtin_amdgcn_alignbit(word >> ((y >> 15) & 31u), hits, 1u);
                }
                out |= (c0 + j < nch32 ? hits >> 24 : 0u) << (8 * j);
                __builtin_amdgcn_sched_barrier(0); // chunk by chunk: the window state of the batches in flight is live here
            }
        }
        return out;
    };
    // CANDIDATE LIST (ApmVerifyArgs::clist) in front of the rows: region g's batches of 64 entries are dealt statically to the
    // waves g, g + R, g + 2R, ... (R regions; fewer waves than regions: wave w takes regions w, w + W, ... whole) -- the
    // regions of a sieve launch fill evenly (its workgroups walk the text interleaved), so there is nothing to balance
    // A short region is cut into as many batches as it has waves (cfg5: 40 entries for 16 waves): one wave working a
    // dense batch alone walks the longest key list among 64 lanes, a chain of dependent gathers, while the others idle.
    bool cl_on = false;                                  // wave-uniform, like the rest
    uint32_t cl_r = 0, cl_rstep = 0, cl_b = 0, cl_bstep = 1, cl_n = 0, cl_bs = 64;
    if constexpr (!FUSED && !SAMPLED) {
        if (a.clist) {
            const uint32_t R = (uint32_t)a.clist_regions, Wv = (uint32_t)gridDim.x * (uint32_t)(THREADS / 64);
            if (Wv >= R) {
                const uint32_t wpr = Wv / R;
                cl_on = my_wave < wpr * R;
                cl_r = my_wave % R;
                cl_rstep = R; // (one region only)
                cl_b = my_wave / R;
                cl_bstep = wpr;
            } else {
                cl_on = my_wave < R;
                cl_r = my_wave;
                cl_rstep = Wv;
            }
            if (cl_on) {
                cl_n = a.clist_cnt[cl_r];
                if (Wv >= R) {
                    const uint32_t per_wave = (cl_n + cl_bstep - 1u) / cl_bstep;
                    const uint32_t lo = (uint32_t)a.clist_min_batch;
                    cl_bs = per_wave >= 64u ? 64u : (per_wave < lo ? lo : per_wave);
                }
            }
        }
    }
    // the next batch: up to 64 positions (in units of STEP bytes); false once the wave's run is exhausted
    auto next_cand = [&](uint32_t &q, bool &hv) __attribute__((always_inline)) -> bool {
        if constexpr (!FUSED && !SAMPLED) {
            while (cl_on) {
                const uint32_t o = cl_b * cl_bs;
                if (o < cl_n) {
                    const uint32_t nb = cl_n - o < cl_bs ? cl_n - o : cl_bs;
                    hv = (uint32_t)lane < nb;
                    q = hv ? a.clist[(size_t)cl_r * a.clist_cap + o + (uint32_t)lane] : 0u;
                    cl_b += cl_bstep;
                    return true;
                }
                cl_r += cl_rstep;
                cl_b = 0; // (whole regions from here on: cl_bs is 64)
                if (cl_r >= (uint32_t)a.clist_regions) { cl_on = false; break; }
                cl_n = a.clist_cnt[cl_r];
            }
        }
        while (qcount < 64u) {
            if (!__builtin_amdgcn_ballot_w64(hm != 0u)) { // block done: take the prefetched masks of the next one
                if constexpr (FUSED) { // ... or sieve the wave's next block
                    uint32_t b;
                    if constexpr (PREF) {
                        if (!pf_started) {
                            pf_started = true;
                            uint32_t nb;
                            if (it_next(nb)) { pf_b = nb; pf_issue(nb); }
                        }
                        if (pf_b == 0xffffffffu) break;
                        b = pf_b;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pf_sl[j] = apm_pack16(pf_r[j].x, pf_r[j].y, pf_r[j].z, pf_r[j].w, cs);
                        uint32_t nb;
                        if (it_next(nb)) { pf_b = nb; pf_issue(nb); }
                        else pf_b = 0xffffffffu;
                    } else if (!it_next(b)) break;
                    blk = tile0 + b * 4096u;
                    uint32_t nblk = 1; // the step's blocks: neighbours out of the same chunk
                    for (; nblk < NBLK && it_b < it_end; ++nblk) ++it_b;
                    hm = sieve_block(blk, b, nblk);
                    continue;
                }
                if (hb_q[0] == NONE) break;
                blk = (uint32_t)(a.tile0 + (int64_t)hb_q[0] * 4096);
                hm = hm_q[0];
#pragma unroll
                for (int i = 0; i + 1 < AHEAD; ++i) { hm_q[i] = hm_q[i + 1]; hb_q[i] = hb_q[i + 1]; }
                {
                    uint32_t b = NONE;
                    hb_q[AHEAD - 1] = it_next(b) ? (listed ? a.blist[b] : b) : NONE;
                    hm_q[AHEAD - 1] = hb_q[AHEAD - 1] != NONE ? a.masks[(uint64_t)hb_q[AHEAD - 1] * 64 + (uint64_t)lane] : 0u;
                }
                continue;
            }
            const bool has = hm != 0u;
            const uint32_t t = has ? (uint32_t)__builtin_ctz(hm) : 0u;
            hm &= hm - 1u;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(has);
            const uint32_t idx = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (has) s_q[idx] = (blk + (t >> 3) * 1024u + 16u * (uint32_t)lane + (SAMPLED ? (t & 1u) * 8u + ((t >> 1) & 3u) * 4096u : (t & 7u) * 2u)) / STEP;
            qcount += (uint32_t)__builtin_popcountll(mask);
        }
        q = 0;
        hv = false;
        if (qcount == 0u) return false;
        const uint32_t nb = qcount < 64u ? qcount : 64u;
        hv = (uint32_t)lane < nb;
        if (hv) q = s_q[lane];
        if (qcount > 64u) { // keep the rest for the next batch
            const uint32_t rest = s_q[64 + lane];
            if ((uint32_t)lane < qcount - 64u) s_q[lane] = rest;
        }
        qcount -= nb;
        return true;
    };
    // one batch ahead: the positions of batch b+1 are formed (a matter of registers and LDS) and its text windows are in
    // flight while batch b is worked on (vmcnt counts in order: the loads of the pre-check queue behind them and wait
    // for no more)
    bool done = false, active = false, have = false;
    uint32_t p = 0, str = 0, s = 0, pend = 0;
    uint32_t cur = 0; // low half: current key id | 0x8000 when it is the last of its list; high half: index of the next slot
    Win win, win_n, wk; // text at the candidate position (this / the next batch); SAMPLED: at the piece the key in hand implies
    constexpr uint32_t PSH = SAMPLED ? 3u : 1u;    // queue entry -> relative position
    constexpr uint32_t KBITS = SAMPLED ? 11u : 15u; // key id bits of a key-list payload; above them the block's offset in its piece
    uint32_t q_n = 0;
    bool have_n = false;
    bool ex_n = true; // (an empty batch starts the pipeline through the loop's own rotate step)
    constexpr int PIPE = FUSED ? APM_FUSED_PIPE : APM_VERIFY_PIPE;
    uint32_t q_nn = 0; // (PIPE == 2)
    bool have_nn = false, ex_nn = true;
    load_win(0u, win_n);
    for (;;) {
        // the DP pass takes FULL waves of (survivor, shift) items only -- what is left over (fewer than 64 items, possibly
        // part of a survivor's shifts: item_lo) waits at the front of the list for the next pass; everything at the end
        if (n_surv * NSH - item_lo >= 64u || (done && n_surv)) {
            const uint32_t avail_items = n_surv * NSH - item_lo, proc = done ? avail_items : (avail_items & ~63u);
            for (uint32_t w0 = 0; w0 < proc; w0 += 64) { // one (survivor, shift) per lane
                const uint32_t wi = item_lo + w0 + (uint32_t)lane;
                const bool live = w0 + (uint32_t)lane < proc;
                const uint2 e = live ? s_surv[wi / NSH] : make_uint2(0u, 0u);
                const int dl = (int)(wi % NSH) - BAND;
                uint32_t wpat = 0, wj = 0, word = 0;
                bool hit = live && core.dp_match(e.y, e.x, dl, wpat, wj, word);
#ifdef APM_MEASURE
                if (APM_SKIP(a, 64)) hit = false;
#endif
                core.count_matches(hit, wpat, wj, word);
            }
            const uint32_t s_first = (item_lo + proc) / NSH, left = n_surv - s_first; // (left <= 22 survivors)
            const uint2 keep = (uint32_t)lane < left ? s_surv[s_first + (uint32_t)lane] : make_uint2(0u, 0u);
            if ((uint32_t)lane < left) s_surv[lane] = keep;
            item_lo = item_lo + proc - s_first * NSH;
            n_surv = left;
        }
        if (done) break;
        // a lane without a key in hand takes up the next of its (at most two) hit positions: key list by rank
        if (!active && pend) {
            const uint32_t par = (pend & 1u) ? 0u : 1u;
            pend &= pend - 1u;
            const uint32_t x = (str >> (2u * par)) & 0xffffu, bit = x >> 11;
            const uint32_t word = s_bmp[x & 2047u];
            const uint32_t e = s_r2s[(uint32_t)s_prefix[x & 2047u] + (uint32_t)__builtin_popcount(word & ((1u << bit) - 1u))];
            s = p + par;
            if (e & 0x8000u) cur = e;
            else cur = (uint32_t)s_slots[e] | ((e + 1u) << 16);
            active = true;
            if constexpr (SAMPLED) { // the block at p lies r bytes inside its piece: the unit's position is p - r
                s = p - ((cur & 0x7fffu) >> KBITS);
                if (s > p) s = 0xffffffffu; // (in front of the shard: the pre-check's position test rejects it)
                load_win(s & ~3u, wk);
            }
        }
        if (!__builtin_amdgcn_ballot_w64(active)) { // this batch is exhausted: rotate the pipeline
            if (!ex_n) { done = true; continue; }
            p = q_n << PSH; // relative position (even / a multiple of 8)
            have = have_n;
            win = win_n;
            if constexpr (PIPE == 2) {
                ex_n = ex_nn;
                q_n = q_nn;
                have_n = have_nn;
                load_win((q_n << PSH) & ~3u, win_n);
                ex_nn = next_cand(q_nn, have_nn);
            } else {
                ex_n = next_cand(q_n, have_n);
                load_win((q_n << PSH) & ~3u, win_n);
            }
            // code words of the 8-byte windows at p and p + 1 (16 bits each) out of the 12 bytes from p on
            str = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const uint32_t b4 = __builtin_amdgcn_alignbyte(win.w[i + 1], win.w[i], p & 3u);
                str |= apm_udot4((b4 >> cs) & 0x03030303u, 0x40100401u) << (8 * i);
            }
            const uint32_t x0 = str & 0xffffu, x1 = (str >> 2) & 0xffffu;
            pend = have ? (((s_bmp[x0 & 2047u] >> (x0 >> 11)) & 1u) | (SAMPLED ? 0u : (((s_bmp[x1 & 2047u] >> (x1 >> 11)) & 1u) << 1))) : 0u;
#ifdef APM_MEASURE
            if (APM_SKIP(a, 8)) pend = 0;
#endif
            continue;
        }
        bool ok = false;
        constexpr uint32_t KMASK = (1u << KBITS) - 1u;
        if (active) ok = core.stage1(cur & KMASK, s, SAMPLED ? wk : win, load_win);
#ifdef APM_MEASURE
        if (APM_SKIP(a, 16)) ok = false;
#endif
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
#ifdef APM_MEASURE
        if (APM_SKIP(a, 256) && mask && lane == 0) atomicAdd(&a.stats[1], (unsigned long long)__builtin_popcountll(mask));
#endif
        if (mask) { // survivors -> the wave's list (ballot + mbcnt, no atomics); at most FLUSH_AT - 1 + 64 entries
            const uint32_t idx = n_surv + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (ok) s_surv[idx] = make_uint2(s, cur & KMASK);
            n_surv += (uint32_t)__builtin_popcountll(mask);
        }
        if (active) {
            if (cur & 0x8000u) active = false;
            else {
                cur = (uint32_t)s_slots[cur >> 16] | ((cur & 0xffff0000u) + 0x10000u);
                if constexpr (SAMPLED) {
                    s = p - ((cur & 0x7fffu) >> KBITS);
                    if (s > p) s = 0xffffffffu; // (in front of the shard: the pre-check's position test rejects it)
                    load_win(s & ~3u, wk);
                }
            }
        }
    }

#ifdef APM_MEASURE
    if (APM_SKIP(a, 512) && lane == 0 && my_wave < APM_STATS_WAVES) a.stats[9 + 2 * my_wave] = wall_clock64();
#endif
    __syncthreads();
    for (int i = tid; i < a.n_pats; i += THREADS) {
        const uint32_t cnt = s_cnt[i];
        if (cnt) atomicAdd(&a.counts[a.pats[i].index], (unsigned long long)cnt);
    }
}

template <int BAND, int THREADS, bool SAMPLED>
__global__ __launch_bounds__(THREADS, (BAND == 1 && THREADS == 256 && !SAMPLED) ? 6 : 4) void apm_verify_kernel(ApmVerifyArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    apm_verify_body<BAND, THREADS, SAMPLED, false>(a, nullptr, smem);
}

static size_t apm_verify_lds_bytes_t(const ApmVerifyArgs &a, int threads) {
    return (size_t)a.image_len + (size_t)((a.n_pats + 3) & ~3) * 4 + (size_t)(threads / 64) * ((size_t)apm_verify_scap(a.band) * 8 + 128 * 4) +
           16; // image + counts + one survivor list and one hit queue per wave
}

static const void *apm_verify_fn(int band, int threads, int stride) {
    const bool big = threads == 512;
    if (stride == 8) {
        switch (band) {
        case 0: return big ? (const void *)apm_verify_kernel<0, 512, true> : (const void *)apm_verify_kernel<0, 256, true>;
        case 1: return big ? (const void *)apm_verify_kernel<1, 512, true> : (const void *)apm_verify_kernel<1, 256, true>;
        case 2: return big ? (const void *)apm_verify_kernel<2, 512, true> : (const void *)apm_verify_kernel<2, 256, true>;
        case 3: return big ? (const void *)apm_verify_kernel<3, 512, true> : (const void *)apm_verify_kernel<3, 256, true>;
        default: return nullptr;
        }
    }
    switch (band) {
    case 0: return big ? (const void *)apm_verify_kernel<0, 512, false> : (const void *)apm_verify_kernel<0, 256, false>;
    case 1: return big ? (const void *)apm_verify_kernel<1, 512, false> : (const void *)apm_verify_kernel<1, 256, false>;
    case 2: return big ? (const void *)apm_verify_kernel<2, 512, false> : (const void *)apm_verify_kernel<2, 256, false>;
    case 3: return big ? (const void *)apm_verify_kernel<3, 512, false> : (const void *)apm_verify_kernel<3, 256, false>;
    default: return nullptr;
    }
}

// workgroup size (256 or 512 threads) and workgroups per CU that put the most waves on a CU for this LDS image
int apm_verify_geometry(const ApmVerifyArgs &a, int *threads) {
    int best_waves = 0, best_blocks = 2;
    *threads = 256;
    for (int t : {256, 512}) {
        int per_cu = 0;
        const void *fn = apm_verify_fn(a.band, t, a.stride);
        if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, t, apm_verify_lds_bytes_t(a, t)) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            continue;
        }
        per_cu = per_cu > 8 ? 8 : per_cu;
        if (per_cu * (t / 64) > best_waves) {
            best_waves = per_cu * (t / 64);
            best_blocks = per_cu;
            *threads = t;
        }
    }
    return best_blocks;
}

hipError_t apm_launch_verify(const ApmVerifyArgs &a, int threads, int max_blocks, int *work_epoch, hipStream_t s) {
    if (a.n_pats <= 0) return hipSuccess;
    const void *fn = apm_verify_fn(a.band, threads, a.stride);
    if (!fn) return hipErrorInvalidValue;
    ApmVerifyArgs args = a;
    args.n_blocks = max_blocks < 1 ? 1 : max_blocks; // (number of hits unknown on the host: a persistent grid shares the blocks)
#ifdef APM_MEASURE
    if (const char *e = getenv("APM_VERIFY_GRID_PCT")) args.n_blocks = std::max(1, (int)((long)args.n_blocks * atoi(e) / 100)); // occupancy sensitivity
    if (const char *e = getenv("APM_MEASURE_SKIP")) args.skip_mask = atoi(e);
#endif
    args.work_groups = std::min<long>(APM_WORK_GROUPS, (long)args.n_blocks * (threads / 64));
    args.work_epoch = *work_epoch;
    void *kargs[] = {&args};
    const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)args.n_blocks), dim3((unsigned)threads), kargs, apm_verify_lds_bytes_t(a, threads), s);
    if (e == hipSuccess) ++*work_epoch; // only a launch that runs advances it: launch e zeroes the counter set launch e + 1 uses
    return e;
}

// ---------------------------------------------------------------------------
// FUSED: sieve + verify in one launch (apm_verify_body<.., FUSED = true>; see ApmFusedArgs)
// ---------------------------------------------------------------------------
#ifndef APM_FUSED_S_WAVES
#define APM_FUSED_S_WAVES 6 /* waves per SIMD the sampled fused form with band 1 is compiled for (80 registers; 5: 86 registers, cfg4 0.205 -> 0.198 ms; 7 and 8 fit only without the prefetch and measured 0.200 / 0.210: profiles/r03/cfg4_ab.txt) */
#endif
template <int BAND, bool SAMPLED>
__global__ __launch_bounds__(APM_FUSED_MAX_THREADS, SAMPLED ? (BAND == 0 ? 7 : (BAND == 1 ? APM_FUSED_S_WAVES : 5)) : (BAND == 0 ? 6 : 5)) void apm_fused_kernel(ApmFusedArgs f) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if ((int)blockIdx.x >= f.s.n_main_blocks) { // extra workgroups: truncated tail windows (one pattern each)
        apm_tail_body(f.s.tail, (int)blockIdx.x - f.s.n_main_blocks, reinterpret_cast<uint4 *>(smem), (int)threadIdx.x);
        return;
    }
    apm_verify_body<BAND, 0, SAMPLED, true>(f.v, &f.s, smem);
}

static const void *apm_fused_fn(int band, int stride) {
    switch (band) {
    case 0: return stride == 8 ? (const void *)apm_fused_kernel<0, true> : (const void *)apm_fused_kernel<0, false>;
    case 1: return stride == 8 ? (const void *)apm_fused_kernel<1, true> : (const void *)apm_fused_kernel<1, false>;
    case 2: return stride == 8 ? (const void *)apm_fused_kernel<2, true> : (const void *)apm_fused_kernel<2, false>;
    case 3: return stride == 8 ? (const void *)apm_fused_kernel<3, true> : (const void *)apm_fused_kernel<3, false>;
    default: return nullptr;
    }
}

size_t apm_fused_lds_bytes(const ApmFusedArgs &a, int threads) {
    const size_t need = (a.s.stride == 8 ? 0 : 32768) + apm_verify_lds_bytes_t(a.v, threads);
    return need < 4096 + 256 ? 4096 + 256 : need; // (the tail workgroups' tables)
}

// workgroup size (a multiple of 64, <= APM_FUSED_MAX_THREADS) and workgroups per CU that put the most waves on a CU
int apm_fused_geometry(const ApmFusedArgs &a, int *threads) {
    const void *fn = apm_fused_fn(a.v.band, a.s.stride);
    int best_waves = 0, best_blocks = 0;
    *threads = 0;
    if (!fn) return 0;
    apm_ensure_max_lds(fn);
#ifdef APM_MEASURE
    static const int forced = getenv("APM_FUSED_THREADS") ? atoi(getenv("APM_FUSED_THREADS")) : 0;
#else
    constexpr int forced = 0;
#endif
    for (int t = APM_FUSED_MAX_THREADS; t >= 256; t -= 64) {
        if (forced && t != forced) continue;
        const size_t lds = apm_fused_lds_bytes(a, t);
        if (lds > (size_t)160 * 1024) continue;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, t, lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            continue;
        }
        if (per_cu * (t / 64) > best_waves) { // (ties: the bigger workgroup, fewer copies of the tables)
            best_waves = per_cu * (t / 64);
            best_blocks = per_cu;
            *threads = t;
        }
    }
    return best_blocks;
}

hipError_t apm_launch_fused(const ApmFusedArgs &a, int threads, int max_blocks, int *work_epoch, hipStream_t s) {
    if (a.s.nchunks <= 0 || a.v.n_pats <= 0) return hipSuccess;
    const void *fn = apm_fused_fn(a.v.band, a.s.stride);
    if (!fn || threads < 64 || threads > APM_FUSED_MAX_THREADS || (threads & 63)) return hipErrorInvalidValue;
    const size_t lds = apm_fused_lds_bytes(a, threads);
    const int64_t n_fb = (a.s.nchunks + 3) / 4, want = (n_fb + threads / 64 - 1) / (threads / 64);
    const int64_t nb = want < max_blocks ? want : (max_blocks < 1 ? 1 : max_blocks);
    ApmFusedArgs args = a;
    args.s.n_main_blocks = (int)nb;
    args.v.n_blocks = (int)nb;
#ifdef APM_MEASURE
    if (const char *e = getenv("APM_MEASURE_SKIP")) args.v.skip_mask = atoi(e);
#endif
    if (lds > 48 * 1024) apm_ensure_max_lds(fn); // (per device: the geometry query ran on one)
    args.v.work_groups = (int)std::min<int64_t>(APM_WORK_GROUPS, nb * (threads / 64));
    args.v.work_epoch = *work_epoch;
    void *kargs[] = {&args};
    const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)(nb + a.s.n_tail)), dim3((unsigned)threads), kargs, lds, s);
    if (e == hipSuccess) ++*work_epoch; // (as in apm_launch_verify)
    return e;
}
