// xv_frontend.hip -- feature front-end of the extraction path on the MI355X.
//
// The reference feeds extract_embedding.py through two Kaldi binaries (local/tf/extract_xvectors.sh:68):
//     apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 scp:feats.scp ark:- |
//     select-voiced-frames ark:- scp,s,cs:vad.scp ark:- |
// Kaldi itself is an external dependency of the recipe (not vendored, version unpinned), so this restates the published
// algorithm of its SlidingWindowCmn (feature-functions.cc): for frame t of an utterance of T frames
//     center:      ws = t - window/2,  we = ws + window          not centered:  ws = t - window, we = t + 1
//     ws < 0  ->   we -= ws, ws = 0                              (not centered: if we > t, we = max(t+1, min_window))
//     we > T  ->   ws -= we - T, we = T, ws = max(ws, 0)
//     out[t]  =    float( double(x[t]) - sum_{ws <= s < we} double(x[s]) / (we - ws) )      (no variance normalisation)
// and select-voiced-frames as a scatter: the host hands dst_row[t] = destination row of input frame t in the packed
// batch (chunk layout of the frames that survive the VAD), or -1 for dropped frames.
//
// HBM-bound, tiny next to the network (92 B/frame in, 96 B/frame out): one 32-lane group per (utterance, 64-frame
// segment) slides a double-precision window sum over its segment (window/64 + 2 reads per frame, all L2 hits), lanes =
// feature dimensions so a row is one coalesced 96-B access.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "xvector_hip.h"

extern "C" void xv_internal_set_error(const char *msg);

namespace {

constexpr int SEG = 64;            // frames per lane group
constexpr int GROUPS = 8;          // lane groups (of 32) per 256-thread block

__global__ __launch_bounds__(256) void cmn_sliding_scatter_kernel(const float *__restrict__ x, int ldx, int F,
                                                                  const int *__restrict__ utt_start,
                                                                  const int *__restrict__ utt_len, int window, int center,
                                                                  int min_window, const int *__restrict__ dst_row,
                                                                  float *__restrict__ y, int ldy)
{
    const int u = blockIdx.y;
    const int T = utt_len[u];
    const int seg = blockIdx.x * GROUPS + (threadIdx.x >> 5);
    const int a = seg * SEG;
    if (a >= T) return;
    const int b = min(a + SEG, T);
    const int f0 = threadIdx.x & 31;
    const long base = utt_start[u];
    auto bounds = [&](int t, int &ws, int &we) {
        if (center) { ws = t - window / 2; we = ws + window; }
        else { ws = t - window; we = t + 1; }
        if (ws < 0) { we -= ws; ws = 0; }
        if (!center && we > t) we = max(t + 1, min_window);
        if (we > T) { ws -= we - T; we = T; if (ws < 0) ws = 0; }
    };
    for (int f = f0; f < F; f += 32) {                 // F <= 32 in every recipe: one trip
        const float *col = x + base * ldx + f;
        int ws, we;
        bounds(a, ws, we);
        double sum = 0.0;
        for (int s = ws; s < we; ++s) sum += (double)col[(long)s * ldx];
        for (int t = a; t < b; ++t) {
            int nws, nwe;
            bounds(t, nws, nwe);
            // the window only ever moves right, by at most one frame at each end
            for (; ws < nws; ++ws) sum -= (double)col[(long)ws * ldx];
            for (; we < nwe; ++we) sum += (double)col[(long)we * ldx];
            const int dst = dst_row[base + t];
            if (dst >= 0) y[(long)dst * ldy + f] = (float)((double)col[(long)t * ldx] - sum / (double)(we - ws));
        }
    }
}

}  // namespace

extern "C" int xv_cmn_sliding_scatter_f32(const float *x, int ldx, int feat_dim, const int32_t *utt_start, const int32_t *utt_len,
                                          int n_utts, int max_len, int cmn_window, int center, int min_window,
                                          const int32_t *dst_row, float *y, int ldy, void *stream)
{
    if (n_utts <= 0 || max_len <= 0) return 0;
    if (!x || !utt_start || !utt_len || !dst_row || !y || feat_dim <= 0 || ldx < feat_dim || ldy < feat_dim || cmn_window <= 0 ||
        min_window <= 0) {
        xv_internal_set_error("cmn_sliding_scatter: bad argument");
        return XV_ERR_BAD_ARG;
    }
    const int gx = (max_len + SEG * GROUPS - 1) / (SEG * GROUPS);
    for (int u0 = 0; u0 < n_utts; u0 += 65535) {
        const int nu = n_utts - u0 < 65535 ? n_utts - u0 : 65535;
        hipLaunchKernelGGL(cmn_sliding_scatter_kernel, dim3(gx, nu), dim3(256), 0, (hipStream_t)stream, x, ldx, feat_dim,
                           utt_start + u0, utt_len + u0, cmn_window, center, min_window, dst_row, y, ldy);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            char buf[256];
            snprintf(buf, sizeof(buf), "cmn_sliding_scatter_kernel: %s", hipGetErrorString(e));
            xv_internal_set_error(buf);
            return (int)e;
        }
    }
    return 0;
}
