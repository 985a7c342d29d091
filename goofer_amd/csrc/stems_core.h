// Shared pieces of the frame walkers (stems.hip: registers for the overlap-add, hop == n_fft / 4; round 5's stems_ring.hip, removed: an LDS
// ring, n_fft 2048 with any even hop): kernel arguments read where they are used, and the per-frame records of 64 frames.
#pragma once
#include "binops_core.h"

typedef float2 __attribute__((aligned(4))) float2_u;   // stems start at arbitrary sample offsets: pair stores are 4-byte aligned

// Read a kernel argument from the kernarg segment at the point of use.  The walkers keep ~40 scalars of wave state across
// their frame loop; arguments that are only needed every 64 frames (the frame-record arrays) or once per note kept live
// beside them pushed the compiler past the 102 SGPRs of a wave, and every overflow costs a v_writelane / v_readlane pair in
// the loop (the noise walker carried 105 such spills).  The empty asm hides the segment pointer from the optimiser, so the
// load cannot be hoisted back to the kernel entry; it is a scalar load from the constant cache.
template <typename T>
__device__ __forceinline__ T cold_arg(size_t offset)
{
    const char __attribute__((address_space(4))) *ka = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *reinterpret_cast<const T __attribute__((address_space(4))) *>(ka + offset);
}
#define COLD(type, field) cold_arg<decltype(type::field)>(offsetof(type, field))

// Per-frame records of 64 consecutive frames, one frame per lane: which note, where in it, which envelope row, the frame's
// picks of f0 and the voicing mask.  A wave reads the record of its current frame with v_readlane — no memory access and
// no chain of dependent scalar loads per frame (frame -> note -> offsets); the block is refilled every 64 frames.
struct frame_block {
    int note, t, T, n, src, base_lo, base_hi;
    float f0, mk;
    uint32_t ny_u;             // noise walker, Philox mode: the random word of the frame's Nyquist bin (see load_nyquist)
    int64_t blk0;

    __device__ __forceinline__ void load(int64_t first, int64_t total_frames, const int *__restrict__ frame_note,
                                         const int64_t *__restrict__ frame_off, const int64_t *__restrict__ sample_off,
                                         const int64_t *__restrict__ row_src, const float2 *__restrict__ picks, int lane)
    {
        blk0 = first;
        int64_t f = first + lane;
        f = f < total_frames ? f : total_frames - 1;
        note = frame_note[f];
        const int64_t fo = frame_off[note], so = sample_off[note];
        t = (int)(f - fo);
        T = (int)(frame_off[note + 1] - fo);
        n = (int)(sample_off[note + 1] - so);
        base_lo = (int)(uint32_t)so;
        base_hi = (int)(so >> 32);
        src = row_src ? (int)row_src[f] : (int)f;
        const float2 pv = picks[f];                           // x[::hop] edge-padded to the frame count (GOOFER.py:1104-1106)
        f0 = pv.x;
        mk = pv.y;
    }
    // Bin M (Nyquist) is the one bin past the 8 x 64 a wave holds, and only lane 0 owns it: drawing its phase inside the frame
    // loop costs a whole Philox block per frame for one lane's word.  Here the 64 frames of the block draw theirs at once, one
    // frame per lane — the same block, word and half philox_u16(.., bin M) names.
    __device__ __forceinline__ void load_nyquist(const goofer_note_params *__restrict__ params, uint64_t seed, int m_bin)
    {
        const uint64_t key = seed ^ ((uint64_t)params[note].seed[0] | ((uint64_t)params[note].seed[1] << 32));
        ny_u = philox_u16(key, (uint64_t)t, (uint32_t)m_bin);
    }
    // Harmonic walker behind goofer_render_batch: the assembly's gather kernel wrote warped copies only for the notes that warp
    // (note_warps, binops_core.h); the frames of the others read the assembled row itself.  Bit 31 of `src` says which.
    __device__ __forceinline__ void mark_plain(const goofer_note_params *__restrict__ params, bool have_formants)
    {
        if (!note_warps(params[note], have_formants)) src |= (int)0x80000000;
    }
    __device__ __forceinline__ bool holds(int64_t f) const { return f >= blk0 && f < blk0 + WAVE; }
};
#define FB_GET(fb, field, idx) __builtin_amdgcn_readlane((fb).field, (idx))
#define FB_GETF(fb, field, idx) __int_as_float(__builtin_amdgcn_readlane(__float_as_int((fb).field), (idx)))
#define FB_BASE(fb, idx) ((int64_t)(((uint64_t)(uint32_t)FB_GET(fb, base_hi, idx) << 32) | (uint32_t)FB_GET(fb, base_lo, idx)))
