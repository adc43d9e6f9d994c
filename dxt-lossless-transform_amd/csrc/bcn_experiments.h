// bcn_experiments.h -- EXPERIMENTS side build only (-DDXTLT_EXPERIMENTS; included at the end of bcn_device.h, never compiled into the shipped
// library): the first form of the forward shifted tiles (typed partial segments; rounds 1-2) and the element-granular kernel (one lane
// per block; the tails and misaligned buffers of rounds 1-3).  Both were replaced by the halo / edge tiles; they stay buildable because the
// measurements in profiles/ that led there are only reproducible with them (tools/shift_probe.py, switches 0x400 / 1 / 0x2000).
#pragma once
#ifndef DXTLT_EXPERIMENTS
#error "bcn_experiments.h belongs to the -DDXTLT_EXPERIMENTS side build"
#endif

namespace dxtlt {

// For image byte o (a multiple of 16): stream index, segment number within the slice, LDS address of the segment, global
// offset of the segment (aligned), the stream's shift and the number of whole segments of the slice.
template <int FMT, bool SA, bool SC, int T>
__device__ __forceinline__ void shifted_segment(int o, const uint64_t (&gb)[6], const Shifts& sh, int& s_out, int& k_out,
                                                int& lds_addr, uint64_t& g_off, int& shift, int& nseg)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    s_out = 0; k_out = 0; lds_addr = 0; g_off = 0; shift = 0; nseg = 0;
#pragma unroll
    for (int s = 0; s < S.n; ++s) {
        const int lo = S.off[s] * T;
        const int hi = lo + S.width[s] * T;
        if (o >= lo && o < hi) {
            s_out = s;
            k_out = (o - lo) >> 4;
            lds_addr = lo + 16 * s + (o - lo);
            g_off = gb[s] + (uint64_t)(o - lo);
            shift = sh.d[s];
            nseg = S.width[s] * T / 16;
        }
    }
}


// EXPERIMENTS build only: the FIRST form of the forward shifted tiles (typed partial segments), kept for the measurements that
// led to the halo tiles below (tools/shift_probe.py, switch 0x400).
// one shifted tile, forward; `lds` is the workgroup's shift_lds_bytes(R) scratch (R = 1: shared with the batch kernel)
template <int FMT, int VARIANT, bool SA, bool SC, int NORM = kNormNone, int R = 1>
__device__ __forceinline__ void fwd_shift_tile(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa,
                                               uint64_t /*total_blocks: in sh.gbase*/, uint64_t /*first_block: in sh.gbase*/, const Shifts& sh, uint64_t tile,
                                               uint8_t* lds)
{
    constexpr Streams S = make_streams(FMT, SA, SC);
    constexpr int T = tile_blocks(FMT, 256) * R;
    const int t = threadIdx.x;
    int base[6];
#pragma unroll
    for (int s = 0; s < 6; ++s)
        base[s] = s < S.n ? S.off[s] * T + 16 * s + sh.d[s] : 0;

    u32x4 q[R];
#pragma unroll
    for (int j = 0; j < R; ++j)
        q[j] = gload16(aos + tile * (4096 * R) + (t + 256 * j) * 16);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const u32x4 v = normalize_vector<FMT, NORM>(q[j]);
        if (sh.natural)
            scatter_shifted<FMT, VARIANT, SA, SC, true>(lds, t + 256 * j, v, base);
        else
            scatter_shifted<FMT, VARIANT, SA, SC, false>(lds, t + 256 * j, v, base);
    }
    __syncthreads();

    uint64_t gb[6];
    slice_bases<FMT, SA, SC, T>(tile, sh, gb);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        int s, k, la, shift, nseg;
        uint64_t g;
        shifted_segment<FMT, SA, SC, T>((t + 256 * j) * 16, gb, sh, s, k, la, g, shift, nseg);
        if (k == 0 && shift > 0) {
            if (!sh.skip_partial)
                copy_partial_segment<true>(soa + g, lds + la, shift, 16);   // (one LDS read + stores from registers: 0.68 against 0.705)
        } else {
            // A 128-byte line that this wave instruction writes completely may use the write-through streaming store of
            // the aligned kernels.  A line that is completed by another wave or by the neighbouring tile must stay in
            // L2 until then: plain `nt` (write-through on those: 0.39-0.50 of peak instead of 0.72-0.76).
            // In segment numbers of the slice: p = this segment's place in its line, kf = the line's first segment.
            // The line lies inside the slice when kf is not the partial head and kf + 8 segments are whole ones; its
            // eight lanes sit in one wave when the first of them is at most lane 56 of the wave.  (Rotating the lanes
            // of a stream so that groups of eight coincide with lines, which makes nearly every line whole, changed
            // nothing: 0.711 against 0.714, profiles/r01_z/shift_probe_rotation_and_chunks.txt.)
            const int p = (int)(((uint32_t)(reinterpret_cast<uintptr_t>(soa) + g) >> 4) & 7u);   // low address bits are enough
            const int kf = k - p;
            const bool whole_line = kf >= (shift > 0 ? 1 : 0) && kf + 8 <= nseg && ((t - p) & 63) <= 56;
            if (sh.line_policy && whole_line)
                gstore16(soa + g, lds_at<u32x4>(lds, la));
            else if (sh.line_policy == 2)
                *reinterpret_cast<u32x4*>(soa + g) = lds_at<u32x4>(lds, la);
            else
                __builtin_nontemporal_store(lds_at<u32x4>(lds, la), reinterpret_cast<u32x4*>(soa + g));
        }
    }
    if (t < S.n && !sh.skip_partial) {  // the extra, partial last segment of stream t
#pragma unroll
        for (int ss = 0; ss < S.n; ++ss) {
            if (ss == t && sh.d[ss] > 0) {
                const int bytes = S.width[ss] * T;
                copy_partial_segment<true>(soa + gb[ss] + bytes, lds + S.off[ss] * T + 16 * ss + bytes, 0, sh.d[ss]);
            }
        }
    }
}

template <int FMT, int VARIANT, bool SA, bool SC, int NORM = kNormNone, int R = 1>
__global__ void __launch_bounds__(256)
fwd_tiled_shift(const uint8_t* __restrict__ aos, uint8_t* __restrict__ soa, uint64_t total_blocks, uint64_t first_block,
                Shifts sh)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[shift_lds_bytes(R)];
    const uint64_t tile = sh.xcd_remap ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (uint64_t)blockIdx.x;
    fwd_shift_tile<FMT, VARIANT, SA, SC, NORM, R>(aos, soa, total_blocks, first_block, sh, tile, lds);
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENTS build only -- the element-granular kernel: one lane per block, natural-width or byte accesses, any alignment, any
// block count.  Rounds 1-3 sent ragged tails and misaligned buffers here (0.4-0.7 of peak); the edge tiles replaced it in round 4.
// ------------------------------------------------------------------------------------------------
// block i (0 <= i < count) of the element-granular range
template <int FMT, int VARIANT, bool SA, bool SC, bool INVERSE, int NORM = kNormNone>
__device__ __forceinline__ void generic_block(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                              uint64_t total_blocks, uint64_t first_block, uint64_t local_first,
                                              uint64_t count, uint64_t i)
{
    // AoS side: block (local_first + i) of the range lives at aos + (local_first + i) * BLOCK.
    // SoA side: global block index first_block + local_first + i.
    constexpr int B = fmt_block(FMT);
    if (i >= count)
        return;
    const uint64_t lb = local_first + i;
    const uint64_t gb = first_block + lb;
    const uint64_t N = total_blocks;

    const uint8_t* aos_c = INVERSE ? nullptr : src + lb * B;
    uint8_t* aos_m = INVERSE ? dst + lb * B : nullptr;
    const uint8_t* soa_c = INVERSE ? src : nullptr;
    uint8_t* soa_m = INVERSE ? nullptr : dst;

    const void* aos_any = INVERSE ? (const void*)aos_m : (const void*)aos_c;
    const void* soa_any = INVERSE ? (const void*)soa_c : (const void*)soa_m;
    const bool aos4 = aligned_to(aos_any, 4);  // block size is a multiple of 8, so uniform per launch
    const bool aos2 = aligned_to(aos_any, 2);
    const uintptr_t soa_base = reinterpret_cast<uintptr_t>(soa_any);

    // field values of the block
    uint64_t alpha8 = 0;          // BC2 alpha
    uint32_t a0 = 0, a1 = 0;      // BC3 alpha endpoints
    uint32_t i01 = 0, i23 = 0, i45 = 0;  // BC3 alpha index halfwords
    uint32_t colours = 0, indices = 0;
    constexpr int CO = (FMT == kBc1) ? 0 : 8;   // colour dword offset in block

    if constexpr (!INVERSE) {
        if constexpr (FMT == kBc2)
            alpha8 = load_bytes<4>(aos_c, aos4) | (load_bytes<4>(aos_c + 4, aos4) << 32);
        if constexpr (FMT == kBc3) {
            a0 = aos_c[0];
            a1 = aos_c[1];
            i01 = (uint32_t)load_bytes<2>(aos_c + 2, aos2);
            i23 = (uint32_t)load_bytes<2>(aos_c + 4, aos2);
            i45 = (uint32_t)load_bytes<2>(aos_c + 6, aos2);
        }
        colours = (uint32_t)load_bytes<4>(aos_c + CO, aos4);
        indices = (uint32_t)load_bytes<4>(aos_c + CO + 4, aos4);
        if constexpr (FMT == kBc1 && NORM != kNormNone)
            normalize_bc1_block<NORM>(colours, indices);
        colours = decorrelate2<VARIANT>(colours);
    }

    // stream addresses (byte offsets from the start of the transformed buffer)
    uint64_t o_alpha = 0, o_a1 = 0, o_aidx = 0, o_col, o_c1 = 0, o_idx;
    if constexpr (FMT == kBc1) {
        o_col = SC ? 2 * gb : 4 * gb;
        o_c1 = 2 * N + 2 * gb;
        o_idx = 4 * N + 4 * gb;
    } else {
        if constexpr (FMT == kBc2) {
            o_alpha = 8 * gb;
        } else {
            o_alpha = SA ? gb : 2 * gb;
            o_a1 = N + gb;
            o_aidx = 2 * N + 6 * gb;
        }
        o_col = 8 * N + (SC ? 2 * gb : 4 * gb);
        o_c1 = 10 * N + 2 * gb;
        o_idx = 12 * N + 4 * gb;
    }
    const bool n2 = ((soa_base) & 1) == 0;  // every 2-byte stream element offset is even, so parity = base parity
    // 4-/8-byte alignment of a stream depends on base + off*N (uniform per launch)
    auto al = [&](uint64_t off, int a) { return ((soa_base + off) & (uint64_t)(a - 1)) == 0; };

    if constexpr (!INVERSE) {
        if constexpr (FMT == kBc2)
            store_bytes<8>(soa_m + o_alpha, alpha8, al(o_alpha, 8));
        if constexpr (FMT == kBc3) {
            if constexpr (SA) {
                soa_m[o_alpha] = (uint8_t)a0;
                soa_m[o_a1] = (uint8_t)a1;
            } else {
                store_bytes<2>(soa_m + o_alpha, a0 | (a1 << 8), n2);
            }
            store_bytes<2>(soa_m + o_aidx + 0, i01, al(o_aidx, 2));
            store_bytes<2>(soa_m + o_aidx + 2, i23, al(o_aidx, 2));
            store_bytes<2>(soa_m + o_aidx + 4, i45, al(o_aidx, 2));
        }
        if constexpr (SC) {
            store_bytes<2>(soa_m + o_col, colours & 0xFFFFu, al(o_col, 2));
            store_bytes<2>(soa_m + o_c1, colours >> 16, al(o_c1, 2));
        } else {
            store_bytes<4>(soa_m + o_col, colours, al(o_col, 4));
        }
        store_bytes<4>(soa_m + o_idx, indices, al(o_idx, 4));
    } else {
        if constexpr (FMT == kBc2)
            alpha8 = load_bytes<8>(soa_c + o_alpha, al(o_alpha, 8));
        if constexpr (FMT == kBc3) {
            if constexpr (SA) {
                a0 = soa_c[o_alpha];
                a1 = soa_c[o_a1];
            } else {
                const uint32_t p = (uint32_t)load_bytes<2>(soa_c + o_alpha, n2);
                a0 = p & 0xFF;
                a1 = p >> 8;
            }
            i01 = (uint32_t)load_bytes<2>(soa_c + o_aidx + 0, al(o_aidx, 2));
            i23 = (uint32_t)load_bytes<2>(soa_c + o_aidx + 2, al(o_aidx, 2));
            i45 = (uint32_t)load_bytes<2>(soa_c + o_aidx + 4, al(o_aidx, 2));
        }
        if constexpr (SC)
            colours = (uint32_t)load_bytes<2>(soa_c + o_col, al(o_col, 2)) |
                      ((uint32_t)load_bytes<2>(soa_c + o_c1, al(o_c1, 2)) << 16);
        else
            colours = (uint32_t)load_bytes<4>(soa_c + o_col, al(o_col, 4));
        indices = (uint32_t)load_bytes<4>(soa_c + o_idx, al(o_idx, 4));
        colours = recorrelate2<VARIANT>(colours);

        if constexpr (FMT == kBc2) {
            store_bytes<4>(aos_m, (uint32_t)alpha8, aos4);
            store_bytes<4>(aos_m + 4, (uint32_t)(alpha8 >> 32), aos4);
        }
        if constexpr (FMT == kBc3) {
            aos_m[0] = (uint8_t)a0;
            aos_m[1] = (uint8_t)a1;
            store_bytes<2>(aos_m + 2, i01, aos2);
            store_bytes<2>(aos_m + 4, i23, aos2);
            store_bytes<2>(aos_m + 6, i45, aos2);
        }
        store_bytes<4>(aos_m + CO, colours, aos4);
        store_bytes<4>(aos_m + CO + 4, indices, aos4);
    }
}

template <int FMT, int VARIANT, bool SA, bool SC, bool INVERSE, int NORM = kNormNone>
__global__ void __launch_bounds__(kThreads)
generic_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t total_blocks,
               uint64_t first_block, uint64_t local_first, uint64_t count)
{
    generic_block<FMT, VARIANT, SA, SC, INVERSE, NORM>(src, dst, total_blocks, first_block, local_first, count,
                                                       (uint64_t)blockIdx.x * kThreads + threadIdx.x);
}

}  // namespace dxtlt
