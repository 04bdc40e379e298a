/*
 * pcgmix_hip.h — C ABI of libpcgmix_hip.so, the MI355X (gfx950) implementation of the
 * PCGmix per-batch hot path.
 *
 * This is the drop-in boundary.  The upstream reference is pure Python and
 * has no FFI of its own; each entry point below replaces the O(B*C*T) part of one reference
 * function and is what a ctypes binding inside the reference's augmentations.py would call
 * (INTEGRATION.md shows that binding).  Conventions:
 *
 *   - every pointer documented "device" is a device pointer owned by the caller;
 *   - no entry point allocates device memory, synchronises, or touches the default stream:
 *     kernels are enqueued on `stream` and the call returns;
 *   - return value: 0 (hipSuccess) or a hipError_t; pcgmix_error_string() names it;
 *   - tensors are dense row-major ("contiguous" in torch terms), float32 unless noted;
 *   - `frames` holds cumulative heart-state boundaries per sample,
 *     [0, S1end, sysEnd, S2end, cycleEnd] (dataloader_physionet.py:151-172), as int32.
 *
 * File:line citations are into the upstream repository
 * (Liisjak/PCGmix-A-Data-Augmentation-Method-for-Heart-Sound-Classification-EXTENDED).
 */
#ifndef PCGMIX_HIP_H
#define PCGMIX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque HIP stream handle (same type as hipStream_t; declared here so that a host
 * language binding does not need the HIP headers). */
typedef struct ihipStream_t* pcgmix_stream_t;

#define PCGMIX_ABI_VERSION 17

/* ABI version of the loaded library (== PCGMIX_ABI_VERSION it was built with). */
int pcgmix_abi_version(void);

/* Human-readable name of an error code returned by any entry point. */
const char* pcgmix_error_string(int err);

/* ------------------------------------------------------------------------------------------
 * Constant operator of the magnitude-warp spline.                                  [host]
 *
 * Replaces the per-(sample, channel) construction
 *     CubicSpline(linspace(0, T-1, n_knots), random_warps[b, :, c])
 * in magnitude_warp(), augmentations.py:674-683 (scipy default bc_type 'not-a-knot').  The
 * break points are the same for every (b, c), so knots -> piecewise-cubic coefficients is
 * one constant linear map; this fills
 *     op[0 .. n_knots-1]                              break points, float64, as numpy's
 *                                                     linspace(0, T-1, n_knots) rounds them
 *     op[n_knots + (p*4 + j)*n_knots + i]             d coef[p][j] / d knot[i],
 *                                                     p = piece 0..n_knots-2,
 *                                                     j = 0..3 multiplies (t - brk[p])^(3-j)
 * Pure host arithmetic; no device is touched.  n_knots >= 2, T >= 2.
 * Size of `op` in doubles: pcgmix_spline_operator_size(n_knots).
 */
int pcgmix_spline_operator_size(int n_knots);
int pcgmix_spline_operator_f64(int T, int n_knots, double* op /* host */);

/* ------------------------------------------------------------------------------------------
 * Partner permutation.                                                              [host]
 *
 * Replaces the per-group shuffle of get_same_label_mix_indices / get_same_wav_mix_indices /
 * get_same_dataset_mix_indices (augmentations.py:500-514, 528-540, 542-556) and the '(mixAll)'
 * draw (augmentations.py:883-884):
 *     mix[group] = random.Random(seed).sample(list(group), len(group))
 * with a FRESH generator per group.  CPython's algorithm is restated exactly: MT19937 seeded by
 * init_by_array over the 32-bit words of |seed|, sample()'s pool branch (always taken when
 * k == n), _randbelow by rejection on getrandbits(n.bit_length()).  Integer results are
 * bit-identical to CPython 3.8-3.12 (checked against `random` in tests/test_host_logic.py).
 *
 *   group_id    host, int32 (B): any labelling of the groups, values in [0, n_groups)
 *   mix         host, int64 (B) out: mix[b] = partner of sample b
 */
int pcgmix_partner_permutation_i64(const int32_t* group_id, int B, int n_groups, uint64_t seed,
                                   int64_t* mix);

/* random.Random(seed).uniform(0, 1) — the probability-gate draw (augmentations.py:869-870). */
double pcgmix_py_uniform01(uint64_t seed);

/* random.Random(seed).randint(0, hi) — the '(rand)' placement offset (augmentations.py:307). */
int64_t pcgmix_py_randint0(uint64_t seed, int64_t hi);

/* ------------------------------------------------------------------------------------------
 * numpy's legacy global random stream, restated (host).  csrc/pcgmix_nprand.hip.
 *
 * Replaces, bit for bit, the draws the reference takes from it in every step:
 *     np.random.seed(seed); lam = np.random.beta(alpha, alpha)        augmentations.py:659-666
 *     np.random.normal(loc=1.0, scale=sigma, size=(B, knot+2, C))     augmentations.py:677
 * `mt_state` is numpy's own generator state — struct { uint32_t key[624]; int pos; }, the address
 * np.random.get_bit_generator().ctypes.state_address — or a private copy of that layout; the
 * functions advance it in place, so the global stream ends where the reference leaves it.
 * numpy's Gaussian cache flag lives outside that struct: counts of normals must be even (the
 * cache is then empty before and after, as it is behind np.random.seed).  All return 0, or 1
 * for arguments outside that contract (the caller then draws with numpy).
 */
/* RandomState.seed(int): mt19937_seed / init_genrand. */
int pcgmix_np_seed(void* mt_state, uint32_t seed);
/* RandomState.beta(a, b) for a <= 1 and b <= 1 (Johnk's algorithm, legacy_beta). */
int pcgmix_np_beta(void* mt_state, double a, double b, double* out);
/* RandomState.normal(loc, scale, n) into out (host, n doubles), n even (legacy_gauss pairs). */
int pcgmix_np_normal_fill(void* mt_state, double loc, double scale, long long n, double* out);

/* One step's (lambda, knots), drawn one step ahead.  pcgmix_npdraw_step returns what
 * seed(seed) -> beta(alpha, alpha) -> normal(1, sigma, n) yield: *lam, and *knots -> n doubles in
 * memory the object owns (valid until the next call on it); `np_state` (numpy's state address, or
 * NULL) receives the generator state behind the draws.  With `lookahead` = L in 1..3 the object
 * owns L + 1 worker threads that draw the blocks of seed + 1 .. seed + L (same alpha, sigma, n)
 * while the caller works on this step; *hit (optional) says whether this call found its block
 * ready.  Contract: 0 < alpha <= 1, n even.  One caller thread per object; a forked child draws
 * inline. */
typedef struct pcgmix_npdraw pcgmix_npdraw;
int pcgmix_npdraw_create(pcgmix_npdraw** out, int lookahead);
void pcgmix_npdraw_destroy(pcgmix_npdraw* d);
int pcgmix_npdraw_step(pcgmix_npdraw* d, uint32_t seed, double alpha, double sigma, long long n,
                       void* np_state, double* lam, const double** knots, int* hit);
/* Blocks found ready / drawn inline so far. */
long long pcgmix_npdraw_stats(pcgmix_npdraw* d, long long* misses);

/* Validate `frames` (int64 (B,5), as the reference's loader yields them) against the signal
 * length and pack the per-step index block read by pcgmix_mix_warp_f32 into `out` (host, e.g.
 * pinned staging): int32 frames[B][5] | mix[B] | rand_off[B][4] (if given) | rect[B][4] (if
 * given).  Returns 0, or 1 = boundaries negative / decreasing, 2 = cycle longer than T (the
 * reference mis-slices silently there, augmentations.py:294-304), 3 = partner index out of
 * range.                                                                            [host] */
int pcgmix_pack_plan_i32(const int64_t* frames, const int64_t* mix, const int32_t* rand_off,
                         const int32_t* rect, int B, int T, int32_t* out);

/* ------------------------------------------------------------------------------------------
 * Fused segment-aware splice (+ optional magnitude warp).                        [device]
 *
 * Replaces, for a whole batch in one launch:
 *   mixup_keepdur_multidim_tensors            augmentations.py:289-337   (plain and '(rand)')
 *   mixup_keepdur_multidim_tensors_salopt     augmentations.py:210-287   (blend part; the
 *                                             displacement comes from pcgmix_salopt_disp_f32)
 *   the per-sample loop around them           augmentations.py:909-917 / 969-977
 *   magnitude_warp + its D2H/H2D round trip   augmentations.py:674-683, 924-928
 *   augmentations2d.mixup_keepdur_multidim_tensors  augmentations2d.py:206-221 (call with
 *                                             C = F rows, T = W columns of the spectrogram)
 *   the 2D mask loops                         augmentations2d.py:320-323, 355-358, 393 (zero_rect)
 *
 * For sample b with partner m = mix_idx[b], heart state k = 0..3:
 *     len1 = frames[b][k+1]-frames[b][k],  len2 = frames[m][k+1]-frames[m][k],  n = min(len1,len2)
 *     o    = off ? min(off[b][k], |len1-len2|) : 0
 *     a    = frames[b][k] + (len1 > len2 ? o : 0)        own-side start
 *     s    = frames[m][k] + (len2 > len1 ? o : 0)        partner-side start
 *     y[b,c,a+i] = x[b,c,a+i]*lam + x[m,c,s+i]*(1-lam)   i = 0..n-1   (fp32 mul, mul, add —
 *                                                        never contracted to an FMA)
 *   everywhere else y = x.  (1-lam) is formed in float32.  Elements whose source index would
 *   fall outside [0,T) are left unblended (the reference would raise on such frames).
 * If `knots` is non-NULL each output is then multiplied by the not-a-knot cubic spline through
 * knots[b,:,c] evaluated at t in float64 (scipy's term order c3 + c2*s + c1*s^2 + c0*s^3 with a
 * running power) and rounded once to float32, as `ret[i] = pat * warper` does.
 *
 *   x, y        device, (B, C, T); y must not alias x
 *   frames      device, int32 (B, 5)
 *   mix_idx     device, int32 (B), values in [0, B)
 *   off         device, int32 (B, 4) >= 0, or NULL
 *   knots       device, float64 (B, n_knots, C) exactly as numpy.random.normal fills it, or NULL
 *   spline_op   device, float64 (pcgmix_spline_operator_size(n_knots)); required iff knots
 *   zero_rect   device, int32 (B, 4) = [row0, row1, col0, col1) per sample, or NULL: outputs with
 *               row0 <= c < row1 and col0 <= t < col1 are written as 0 — the masks that
 *               durmixcutout / durmixtimemask / durmixfreqmask apply after the 2D splice
 *               (augmentations2d.py:309-323, 348-358, 384-394)
 */
int pcgmix_mix_warp_f32(const float* x, float* y,
                        const int32_t* frames, const int32_t* mix_idx, const int32_t* off,
                        float lam,
                        const double* knots, const double* spline_op, int n_knots,
                        const int32_t* zero_rect,
                        int B, int C, int T, pcgmix_stream_t stream);

/* The plain splice (no '(rand)' offsets, no warp, no rectangle) with its index block in the KERNEL
 * ARGUMENTS: frames16 (B,5) and mix16 (B) are HOST arrays of int16 that the launch copies into the
 * kernel's 4 KB argument segment (3 KB used) — the blocks read them with scalar loads, and there is
 * no host-to-device copy in front of the launch.  Needs B <= 256, T <= 32767, T % 4 == 0, x and y
 * 16-byte aligned (BASELINE configs[1]); hipErrorInvalidValue otherwise.  Same arithmetic and
 * results as pcgmix_mix_warp_f32.  pcgmix_augment_plain_f32 takes this path by itself.
 * pcgmix_mix_karg_variant: 1 and the instantiation's unroll if (B, C, T) qualifies, else 0.    */
int pcgmix_mix_karg_f32(const float* x, float* y, const int16_t* frames16, const int16_t* mix16,
                        float lam, int B, int C, int T, pcgmix_stream_t stream);
int pcgmix_mix_karg_variant(int B, int C, int T, int* unroll);

/* ------------------------------------------------------------------------------------------
 * Saliency post-processing.                                                       [device]
 *
 * Replaces saliency.get_saliency_maps(), saliency.py:63-91 (dim == 1), after the model's
 * backward pass: |grad| -> zero t >= frames[b][4] -> sum over channels -> `ksize`-tap Gaussian
 * (weights 1/(sigma*sqrt(2*pi)) * exp(-r^2 / (2 sigma^2)), not renormalised, zero 'same'
 * padding, saliency.py:15-18) -> zero the tail again -> per row (s - min) / max(s - min),
 * NaN -> 0.
 *
 *   grad        device, (B, C, T)
 *   sal         device, (B, T) out
 *   ksize       odd, <= 255 (the reference uses 101); weights are formed on the host in
 *               float64 exactly as gaussian_kernel() does and rounded to float32
 *   sigma       float64, the reference's (12/101)*ksize
 *   T           <= 19000 (the row is staged in LDS)
 */
int pcgmix_saliency_post_f32(const float* grad, const int32_t* frames, float* sal,
                             int ksize, double sigma, int B, int C, int T,
                             pcgmix_stream_t stream);

/* The spectrogram branch of the same function (saliency.py:93-113, dim = 2).  grad (B, F, W)
 * float32 (the single image channel folded away), frames (B,5) in spectrogram columns, sal (B, W):
 * |grad| -> columns t >= f[4] zeroed -> sum over the F frequency rows -> `ksize`-tap Gaussian along
 * time (the reference: 11 taps, sigma 1) with zero 'same' padding -> tail zeroed -> min/max
 * normalisation over the cycle's own columns [0, f[4]) ONLY -> NaN -> 0.  W <= 1024.  [device] */
int pcgmix_saliency_post2d_f32(const float* grad, const int32_t* frames, float* sal, int ksize,
                               double sigma, int B, int F, int W, pcgmix_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Saliency-optimal displacement search.                                           [device]
 *
 * Replaces optimal_displacement_max_envelope (mode 0, augmentations.py:60-93) and
 * optimal_displacement_max_sum (mode 1, augmentations.py:95-128) for every (sample, state):
 * disp[b][k] = first strict argmax over d in [0, |len1-len2|] of the float32 objective,
 * evaluated in numpy's pairwise-summation order so that the integer result is the
 * reference's; 0 where the two lengths agree.
 *
 *   sal         device, (B, T) saliency maps
 *   disp        device, int32 (B, 4) out — feeds pcgmix_mix_warp_f32's `off`
 *   workspace   device, pcgmix_salopt_workspace_bytes(B) bytes, 8-byte aligned: the candidates of
 *               one (sample, state) are shared out over several blocks; their partial arg-maxima
 *               meet here (greatest value, smallest displacement on ties = first strict maximum)
 *   max_len     the longest heart state of the batch in samples, max_k,b(frames[b][k+1] -
 *               frames[b][k]) — the caller has `frames` on the host — or 0 for "unknown" (= T).
 *               It sizes the blocks' LDS (2 * max_len floats): the smaller, the more blocks a CU
 *               holds.  States longer than max_len are cut to it (memory safety only).
 *   B <= 65535, T <= 19000
 */
long long pcgmix_salopt_workspace_bytes(int B);
int pcgmix_salopt_disp_f32(const float* sal, const int32_t* frames, const int32_t* mix_idx,
                           float lam, int mode, int32_t* disp, void* workspace, int max_len, int B,
                           int T, pcgmix_stream_t stream);
/* The same search for a caller that ALSO holds the boundaries and the partners on the host — the
 * reference's own situation: mixup_keepdur_multidim_tensors_salopt receives `frames` and the partner
 * index as CPU arrays (augmentations.py:210-287).  frames_host (B,5) / mix_host (B): host copies of
 * `frames` / `mix_idx` (same values: the device copies stay the ones the kernel reads).  With them
 * (B <= 256) the launch holds only the blocks that have candidates, ordered by the length of their
 * chain of sums, longest first, in the kernel arguments; either NULL: as pcgmix_salopt_disp_f32.
 * Results are identical.                                                                        */
int pcgmix_salopt_disp_hosted_f32(const float* sal, const int32_t* frames, const int32_t* mix_idx,
                                  float lam, int mode, int32_t* disp, void* workspace, int max_len,
                                  int B, int T, pcgmix_stream_t stream, const int32_t* frames_host,
                                  const int32_t* mix_host);
/* Host only: that launch plan, for inspection and tests.  ids_out receives up to `cap` block ids
 * ((sample << 4) | (state << 2) | slice: slice z of pair (sample, state) holds the displacements
 * 256 z .. 256 z + 255, + 1024, ...) in launch order — longest chain of sums first; a pair without a
 * search (equal lengths) is represented by its slice 0.  Returns the number of blocks, 0 when no plan
 * is made for this shape (B > 256 or more than 1408 blocks: the full grid is launched), < 0 for
 * bad arguments.                                                                        [host]   */
int pcgmix_salopt_plan(const int32_t* frames_host, const int32_t* mix_host, int B, int T, int max_len,
                       uint16_t* ids_out, int cap);
/* The saliency-guided splice — mixup_keepdur_multidim_tensors_salopt for the whole batch
 * (augmentations.py:210-287 with :60-128, the loop at :909-917, magnitude_warp :674-683) — in one
 * call: the search above, then pcgmix_mix_warp_f32's kernel, whose blocks reduce the search's
 * per-block results for their own sample themselves (no launch in between).  knots / spline_op /
 * n_knots as in pcgmix_mix_warp_f32 (NULL, NULL, 0: no warp).  disp_out: NULL, or int32 (B,4) that
 * receives the displacements from a small launch BEHIND the splice.                            */
int pcgmix_salopt_mix_warp_f32(const float* x, float* y, const float* sal, const int32_t* frames,
                               const int32_t* mix_idx, float lam, int mode, const double* knots,
                               const double* spline_op, int n_knots, void* workspace, int max_len,
                               int32_t* disp_out, int B, int C, int T, pcgmix_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Log-mel front end.                                                    [host tables + device]
 *
 * Replaces the offline librosa pipeline of databuilder.ipynb cell 6:19-23, 81-101, 127-142
 * (melspectrogram -> power_to_db(ref=np.max) -> (x-mean)/std -> slice the cycle's columns ->
 * zero-pad to W columns), on device.  librosa 0.9.2 semantics restated (centred frames, periodic
 * Hann of n_fft, float64 transform rounded to complex64, power 2, Slaney mel scale and
 * normalisation, amin 1e-10, top_db 80); parity with librosa itself is UNPINNED (librosa is not
 * available offline and the reference stores no spectrogram).  One point of that restatement is
 * open and therefore a parameter: `pad_mode` of the centred frames, 0 = zeros (numpy 'constant')
 * or 1 = 'reflect'.
 *
 * Two granularities:
 *   pcgmix_logmel_f32             one transform per heart-cycle item of a batch; `ref` is the
 *                                 item's own maximum (what a per-batch transform of cut cycles
 *                                 can see).
 *   pcgmix_logmel_recordings_f32  the reference's order of operations: ONE transform over each
 *                                 whole recording, `ref` = the recording's maximum (cell 6:93),
 *                                 then cycle c keeps the recording's columns
 *                                 [round(f0*n_frames/len(y)), round(f4*n_frames/len(y)))
 *                                 (cell 6:101, 134), zero-padded to W after normalisation.
 *
 * pcgmix_logmel_tables builds, on the host, everything that does not depend on the data: the
 * windowed DFT matrix in matrix-core operand order, the mel filter bank and each filter's
 * non-zero span (pcgmix_logmel_tables_size bytes).  The caller copies that blob to the device
 * once and passes it to every pcgmix_logmel_f32 call with the same (n_fft, n_mels).
 *
 *   x           device, (B, T) one channel per row
 *   frames      device, int32 (B, 5) waveform boundaries; columns >= round(f4 * n_frames / T)
 *               are zero-filled after normalisation
 *   tables      device, the blob above
 *   spec        device, (B, n_mels, W) out
 *   frames_out  device, int32 (B, 5) out or NULL: boundaries in spectrogram columns
 */
long long pcgmix_logmel_tables_size(int n_fft, int n_mels);
int pcgmix_logmel_tables(int n_fft, int n_mels, float fmin, float fmax, float sr, void* out /* host */);
int pcgmix_logmel_f32(const float* x, const int32_t* frames, const void* tables, float* spec,
                      int32_t* frames_out, int B, int T, int n_fft, int hop, int n_mels,
                      float mean, float std, int W, int pad_mode, pcgmix_stream_t stream);
/* The same for a caller that holds the boundaries on the host (as the reference does: numpy
 * arrays, databuilder.ipynb cell 6:101): frames_host (B,5) int32 HOST memory.  B <= 1024 and
 * T <= 32767: the cycle ends travel in the kernel arguments — no upload, no copy kernel in front
 * of the launch; otherwise hipErrorInvalidValue (upload, then pcgmix_logmel_f32).               */
int pcgmix_logmel_hostframes_f32(const float* x, const int32_t* frames_host, const void* tables,
                                 float* spec, int B, int T, int n_fft, int hop, int n_mels,
                                 float mean, float std, int W, int pad_mode, pcgmix_stream_t stream);

/* Per-recording front end.  All pointers device.  The column bookkeeping (which is integer work
 * on a handful of numbers per cycle: n_frames = 1 + len/hop, Python round()) is the caller's; the
 * kernels take it as descriptor tables:
 *   y            recordings back to back, float32
 *   rec_off      int64 (R): first sample of recording r in y;  rec_len int32 (R): its length
 *   tiles        int32 (n_tiles, 4): {recording, first frame, frames in this tile
 *                (<= pcgmix_logmel_tile_frames()), absolute scratch column of that first frame};
 *                the tiles of a recording cover its frames 0 .. n_frames-1 once
 *   cycles       int32 (n_cycles, 4): {recording, absolute scratch column of the cycle's first
 *                column, number of columns kept (clipped to W), 0}
 *   db_scratch   float32 (n_mels, scratch_cols) workspace: un-referenced dB of every frame
 *   ref_pow      uint32 (R) workspace: bit pattern of the recording's maximum mel power
 *   spec         float32 (n_cycles, n_mels, W) out
 * Three launches: zero ref_pow; transform + mel + dB per tile (f64 matrix cores), maximum folded
 * in with an unsigned atomicMax (non-negative floats order like their bit patterns: the result is
 * order-independent); slice / reference / clip / normalise / pad per cycle.
 */
int pcgmix_logmel_tile_frames(void);
int pcgmix_logmel_recordings_f32(const float* y, const int64_t* rec_off, const int32_t* rec_len,
                                 int R, const int32_t* tiles, int n_tiles, const int32_t* cycles,
                                 int n_cycles, const void* tables, float* db_scratch,
                                 long long scratch_cols, uint32_t* ref_pow, float* spec, int n_fft,
                                 int hop, int n_mels, float mean, float std, int W, int pad_mode,
                                 pcgmix_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Potes 1D-CNN convolutional branch, fused.                                         [device]
 *
 * Replaces, for the N = 4*B band rows of a batch, CNN_potes.cnn1 (models.py:359-381, 444-455):
 *     Conv1d(1->8,k5,pad1) + ReLU + MaxPool1d(2) -> Conv1d(8->4,k5,pad1) + ReLU + MaxPool1d(2)
 * (the Dropout(0.25) that follows is applied by the caller).  Weight layouts are torch's:
 * w1 (8,1,5), b1 (8), w2 (4,8,5), b2 (4).
 *
 *   pcgmix_potes_out_len(T)        pooled output length P2 (1248 for T = 5000, 623 for 2500)
 *   pcgmix_potes_stack_fwd_f32     x (N,T) -> h2 (N,4,P2)
 *   pcgmix_potes_bwd_blocks(N,T)   G = number of partial rows the backward needs
 *   pcgmix_potes_stack_bwd_f32     x, dL/dh2 (N,4,P2) -> grads[212] = [gw1(40) | gb1(8) |
 *                                  gw2(160) | gb2(4)]; `partial` is caller-provided scratch of
 *                                  G*212 floats.  The forward is recomputed tile by tile.
 *   pcgmix_potes_stack_input_grad_f32   x, dL/dh2 -> dL/dx (N,T): what saliency.py:52-61 needs
 *                                  (the class score differentiated w.r.t. the input); also
 *                                  recomputes the forward, keeps nothing from it.
 * All pointers device; float32.  Both forwards run on the f32 matrix cores
 * (v_mfma_f32_4x4x1_16b_f32: every output is an in-order fmaf chain from the bias over (input
 * channel, tap) — exact f32); the recomputing backward kernels and the layer-1 recompute of the
 * mask-based weight gradient use a VALU summation order and agree with it to float rounding.
 */
int pcgmix_potes_out_len(int T);
int pcgmix_potes_bwd_blocks(int N, int T);
int pcgmix_potes_stack_fwd_f32(const float* x, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* h2, int N, int T, pcgmix_stream_t stream);
int pcgmix_potes_stack_input_grad_f32(const float* x, const float* grad_h2, const float* w1,
                                      const float* b1, const float* w2, const float* b2,
                                      float* grad_x, int N, int T, pcgmix_stream_t stream);
int pcgmix_potes_stack_bwd_f32(const float* x, const float* grad_h2, const float* w1,
                               const float* b1, const float* w2, const float* b2, float* partial,
                               float* grads, int N, int T, pcgmix_stream_t stream);
/* The same three passes with the forward's ReLU / max-pool routing SAVED instead of recomputed
 * (what autograd keeps for nn.ReLU / nn.MaxPool1d in models.py:359-365, at 2 bits or a byte per
 * position instead of the activations):
 *   m2   uint8 (N, 4, ceil(P2/4))  second layer: per pooled output 2 bits — 0 ReLU-dead, 1 the
 *        first conv output of the pooled pair won, 2 the second; four outputs per byte
 *   s1   uint8 (N, 8, P1/4 + 1)    first layer: the same selector, four positions per byte
 *        (position q in bits 2*((q+1)&3) of byte (q+1)>>2); only the input gradient needs it
 *        (pass NULL to the forward otherwise)
 *   sizes in bytes: pcgmix_potes_mask_bytes(N, T, layer = 2 or 1)
 *   pcgmix_potes_stack_fwd_save_f32        the forward above + m2 (+ s1)
 *   pcgmix_potes_stack_bwd_mask_f32        weight gradients: recomputes layer 1 only
 *   pcgmix_potes_stack_input_grad_mask_f32 dL/dx from grad_h2, m2, s1 and the weights alone:
 *                                          no forward recompute at all
 * Gradients flow exactly where the forward's maxima were (the recomputing kernels re-derive the
 * routing with a different summation order and can differ at exact near-ties).
 *
 * rnd_out != NULL: the forward launch also fills rnd_out[0 .. rnd_bytes) (16-byte aligned, a
 * multiple of 16) with the uniformly random bytes the head's dropouts read (mask1 / mask2 of
 * pcgmix_potes_head_*): 32-bit word w = a keyed counter hash of w.  The 64-bit key is `key`, or —
 * key_dev != NULL — the two 32-bit words (low, high) at key_dev in DEVICE memory: for training
 * steps captured in a hipGraph, nn.Dropout's mask generation (models.py:364, 380) without an RNG
 * launch per replay — the key changes per step, the graph does not.  rnd_out == NULL: nothing is
 * filled (rnd_bytes, key_dev, key ignored).
 */
long long pcgmix_potes_mask_bytes(int N, int T, int layer);
int pcgmix_potes_stack_fwd_save_f32(const float* x, const float* w1, const float* b1,
                                    const float* w2, const float* b2, float* h2, uint8_t* m2,
                                    uint8_t* s1, int N, int T, uint8_t* rnd_out, long long rnd_bytes,
                                    const uint32_t* key_dev, uint64_t key, pcgmix_stream_t stream);
int pcgmix_potes_stack_bwd_mask_f32(const float* x, const float* grad_h2, const uint8_t* m2,
                                    const float* w1, const float* b1, const float* w2,
                                    const float* b2, float* partial, float* grads, int N, int T,
                                    pcgmix_stream_t stream);
int pcgmix_potes_stack_input_grad_mask_f32(const float* grad_h2, const uint8_t* m2,
                                           const uint8_t* s1, const float* w1, const float* w2,
                                           float* grad_x, int N, int T, pcgmix_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Skinny linear layer forward (the Potes head's `dimreduc`, models.py:376, 430).    [device]
 *
 *   z (B,O) = h (B,K) . W (O,K)^T + bias (O)      O in {8, 16, 20}, K % 4 == 0
 * Split-K with a deterministic two-stage reduction; `partial` is caller-provided scratch of
 * pcgmix_skinny_linear_splits(B,K) * B * O floats.  h and W must be 16-byte aligned.
 */
int pcgmix_skinny_linear_splits(int B, int K);
int pcgmix_skinny_linear_fwd_f32(const float* h, const float* W, const float* bias, float* partial,
                                 float* z, int B, int K, int O, pcgmix_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gradient value clipping + Adam update of one parameter tensor.                    [device]
 *
 * Replaces nn.utils.clip_grad_value_(…, grad_clip) followed by torch.optim.Adam.step() with L2
 * weight decay (train_model.py:557-558, 404-407, 566) for one tensor, in one pass:
 *   g = clamp(g, -clip, clip) (clip <= 0: no clipping) + weight_decay * p
 *   m = m + (1-beta1)(g - m);  v = beta2 v + (1-beta2) g^2
 *   p -= lr / (1-beta1^step) * m / (sqrt(v) / sqrt(1-beta2^step) + eps)          step >= 1
 * p, m, v are updated in place; g is read only.
 */
int pcgmix_adam_clip_f32(float* p, const float* g, float* m, float* v, long long n, float clip,
                         float lr, float beta1, float beta2, float eps, float weight_decay,
                         long long step, pcgmix_stream_t stream);

/* The same update for n_tensors parameter tensors in one launch (one per 32 tensors).  p, g, m,
 * v are HOST arrays of n_tensors device pointers, n the HOST array of element counts; all tensors
 * share the hyper-parameters and the step number (one torch param group).                    */
int pcgmix_adam_clip_multi_f32(int n_tensors, float* const* p, const float* const* g,
                               float* const* m, float* const* v, const long long* n, float clip,
                               float lr, float beta1, float beta2, float eps, float weight_decay,
                               long long step, pcgmix_stream_t stream);

/* The multi-tensor update with its eight scalars in DEVICE memory, for a launch captured in a
 * hipGraph (OneCycleLR moves lr and beta1 every step, the bias corrections move with `step`:
 * kernel arguments would be frozen at capture).  pcgmix_adam_hyper computes the eight floats on
 * the host exactly as pcgmix_adam_clip_multi_f32 does internally — {clip, weight_decay, 1-beta1,
 * beta2, 1-beta2, lr/(1-beta1^step), 1/sqrt(1-beta2^step), eps} — and the caller gets them to
 * hyper_dev before the replay (e.g. with pcgmix_ctx_set_payload).                              */
int pcgmix_adam_hyper(float clip, float lr, float beta1, float beta2, float eps, float weight_decay,
                      long long step, float* out8);
int pcgmix_adam_clip_multi_dev_f32(int n_tensors, float* const* p, const float* const* g,
                                   float* const* m, float* const* v, const long long* n,
                                   const float* hyper_dev, pcgmix_stream_t stream);
/* The same update with the Potes conv stack's gradient reduction folded in (one launch less at
 * the end of a captured training step): pcgmix_potes_stack_bwd_mask_f32 called with grads = NULL
 * leaves its G x 212 partial rows un-reduced; this launch carries 212 extra blocks, each of which
 * sums one column in pcgmix_potes_reduce_f32's order (same bits), stores it at grads[e] and updates
 * the element of the tensor whose gradient pointer g[i] lies inside grads[0 .. 212) (such tensors
 * are updated by those blocks only).  n_tensors <= 32.  pcgmix_potes_reduce_f32 is the reduction
 * alone (what pcgmix_potes_stack_bwd_mask_f32 launches when grads != NULL).                      */
int pcgmix_adam_clip_multi_reduce_dev_f32(int n_tensors, float* const* p, const float* const* g,
                                          float* const* m, float* const* v, const long long* n,
                                          const float* hyper_dev, const float* partial, float* grads,
                                          int G, pcgmix_stream_t stream);
int pcgmix_potes_reduce_f32(const float* partial, float* grads, int G, pcgmix_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Potes classifier head, forward and backward.                                       [device]
 *
 * Replaces CNN_potes' Dropout(.25) -> Flatten/concat -> dimreduc Linear(K->20) -> ReLU ->
 * Dropout(.5) -> Linear(20->C) (models.py:364, 376-381, 456-465) and their autograd twins.
 *   x       (B,K) features BEFORE the Dropout(.25) of the conv branch (K % 4 == 0, 16-byte aligned)
 *   mask1   random bits or NULL (no dropout): element e = b*K + k owns bits1 (1, 2, 4 or 8)
 *           consecutive bits at bit offset e*bits1 and is kept iff their value >= thr1, kept
 *           values times scale1.  bits1 = 8, thr1 = 1 reads a 0/1 byte mask; uniformly random bits
 *           with thr1 = 2^bits1 * p drop with probability p exactly (p = 0.25: 2 bits, thr 1;
 *           scale = 1/(1-p)).  4-byte aligned.  Applied where x is read: no separate pass.
 *   w1,b1   dimreduc (20,K), (20) — b1 may be NULL;   w2,b2  linear (C,20), (C), C <= 8
 *   mask2   (B,20) bytes, kept iff byte >= thr2, NULL = no dropout (eval); scale2 as scale1
 *   partial workspace, pcgmix_skinny_linear_splits(B,K) * B * 20 floats
 *   z       (B,20) out: dimreduc output incl. bias, kept for backward;  logits (B,C) out
 * Backward, given dlogits (B,C) and the SAME x, mask1, mask2:
 *   dz (B,20) workspace/out (gradient at z; 16-byte aligned); dw2 (C,20), db2 (C, may be NULL), db1 (20, may be
 *   NULL), dw1 (20,K; may be NULL when dx is given: frozen weights, e.g. the saliency model — x is
 *   then not read at all), dx (B,K, may be NULL) = gradient at the features BEFORE Dropout(.25).
 * x is read once and dx written once (9 bytes per feature element); reductions are fixed-order.
 */
int pcgmix_potes_head_fwd_f32(const float* x, const uint8_t* mask1, float scale1, int thr1,
                              int bits1, const float* w1, const float* b1, const uint8_t* mask2, float scale2,
                              int thr2, const float* w2, const float* b2, float* partial, float* z,
                              float* logits, int B, int K, int C, pcgmix_stream_t stream);
int pcgmix_potes_head_bwd_f32(const float* dlogits, const float* z, const uint8_t* mask2,
                              float scale2, int thr2, const float* w2, const float* x,
                              const uint8_t* mask1, float scale1, int thr1, int bits1,
                              const float* w1, float* dz, float* dw2, float* db2, float* db1,
                              float* dw1, float* dx,
                              int B, int K, int C, pcgmix_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Potes classifier head + soft-target cross entropy, fused.                           [device]
 *
 * The head above followed by CELoss (train_model.py:45-54: mean_b(-sum_c log_softmax(logits)[b,c]
 * * target[b,c])) as ONE autograd node: the row-local part of the four small kernels between the
 * split-K product and the head's backward (tail forward, CE forward, CE backward, tail backward)
 * runs in one launch, their batch reductions in a second.
 *   forward : x, masks, weights, target (B,C float, soft targets) -> z, logits, loss (scalar) and,
 *             for d loss = 1: dz (B,20) and small = [dW2 (C x 20) | db2 (C) | db1 (20)];
 *             ws: pcgmix_potes_head_loss_workspace_floats(B) floats; dw1_zero (20,K) or NULL: the
 *             buffer the backward accumulates dW1 into, cleared here.
 *   backward: gscale = device scalar d L / d loss (NULL = 1): dx and dW1 from dz * gscale in one pass
 *             over x (as pcgmix_potes_head_bwd_f32), small_out = small_in * gscale.
 */
/* The head of a FROZEN model in eval mode, for d score / d input (saliency.py:52-61): features ->
 * split-K product -> dz = (z > 0) * (seed W2) -> dx = dz W1, three launches, no logits, no weight
 * gradients.  seed (B,C) = d score / d logits (the one-hot of the label; pcgmix_ctx_labels_begin
 * can write it); partial as for pcgmix_potes_head_fwd_f32; dz (B,20) scratch, 16-byte aligned;
 * dx (B,K) out.                                                                              */
int pcgmix_potes_head_saliency_f32(const float* x, const float* w1, const float* b1, const float* w2,
                                   const float* seed, float* partial, float* dz, float* dx, int B,
                                   int K, int C, pcgmix_stream_t stream);
/* The whole saliency pass of a frozen CNN_potes (saliency.py:26-91 with models.py:444-465 as the
 * model) in one call: the six launches of pcgmix_potes_stack_fwd_save_f32 ->
 * pcgmix_potes_head_saliency_f32 -> pcgmix_potes_stack_input_grad_mask_f32 ->
 * pcgmix_saliency_post_f32 back to back.  x (B,4,T); c*: conv stack weights; h*: head weights
 * (hb1 may be NULL); h2, m2, s1, partial, dz, gfeat, gx: the intermediate buffers of those entry
 * points (sizes as documented there); frames (B,5) device; sal (B,T) out.            [device]   */
int pcgmix_potes_saliency_pass_f32(const float* x, const float* cw1, const float* cb1, const float* cw2,
                                   const float* cb2, float* h2, uint8_t* m2, uint8_t* s1,
                                   const float* hw1, const float* hb1, const float* hw2,
                                   const float* seed, float* partial, float* dz, float* gfeat, float* gx,
                                   const int32_t* frames, float* sal, int ksize, double sigma, int B,
                                   int T, int K, int n_classes, pcgmix_stream_t stream);
long long pcgmix_potes_head_loss_workspace_floats(int B);
int pcgmix_potes_head_loss_fwd_f32(const float* x, const uint8_t* mask1, float scale1, int thr1,
                                   int bits1, const float* w1, const float* b1, const uint8_t* mask2,
                                   float scale2, int thr2, const float* w2, const float* b2,
                                   const float* target, float* partial, float* z, float* logits,
                                   float* dz, float* loss, float* small, float* ws, float* dw1_zero,
                                   int defer_finalize, int target_kind, int B, int K, int C,
                                   pcgmix_stream_t stream);
int pcgmix_potes_head_loss_bwd_f32(const float* dz, const float* gscale, const float* x,
                                   const uint8_t* mask1, float scale1, int thr1, int bits1,
                                   const float* w1, const float* small_in, float* small_out,
                                   float* dw1, float* dx, const float* deferred_ws,
                                   float* deferred_loss, int B, int K, int C,
                                   pcgmix_stream_t stream);
/* target_kind: 0 = `target` is float (B,C) (soft targets, CELoss' general case); 1 = `target` points
 * to uint8 class labels (B): hard targets as one byte per row.
 * defer_finalize / deferred_ws, deferred_loss: the forward's last launch — the fixed-order sum of
 * the per-row-block contributions in `ws` into `loss` and `small` — can be left to the backward,
 * whose feature pass does it in its first block (same order, same values): one launch less per
 * step when forward and backward always run together (a training step captured in a hipGraph).
 * With defer_finalize != 0 the forward writes neither `loss` nor `small`; the backward then takes
 * deferred_ws = ws, writes the loss to deferred_loss and ignores small_in.                      */

/* ------------------------------------------------------------------------------------------
 * Soft-target cross entropy (train_model.py:45-54, CELoss).                          [device]
 *   loss[0] = mean_b( -sum_c log_softmax(logits)[b,c] * target[b,c] )
 *   dlogits[b,c] = gout[0] / B * (softmax[b,c] * sum_c' target[b,c'] - target[b,c])
 * logits, target (B,C) float32; gout: device scalar (the incoming gradient of the loss).
 */
int pcgmix_soft_ce_fwd_f32(const float* logits, const float* target, float* loss, int B, int C,
                           pcgmix_stream_t stream);
int pcgmix_soft_ce_bwd_f32(const float* logits, const float* target, const float* gout,
                           float* dlogits, int B, int C, pcgmix_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One-call step of the plain PCGmix methods (durratiomixup / durmixmagwarp, same-label partners,
 * no '(rand)', no saliency): everything augmentations.py:941, 962-977 (and :874, 906-928) do
 * between the probability gate and the returned tensor.                      [host + device]
 *
 *   labels (B) int64, frames (B,5) int64: HOST.  x, y (B,C,T): device.  lam: float32(lambda).
 *   knots: HOST (B,n_knots,C) float64 as numpy drew them, or NULL (no warp); spline_op: DEVICE
 *   operator of pcgmix_spline_operator_f64 for (T, n_knots).
 *   staging: PINNED host scratch, dev_idx: device scratch, both pcgmix_splice_staging_bytes(B, C,
 *   n_knots) bytes; staging may be reused once the copy enqueued here has completed (record an
 *   event on `stream` after the call).
 *   mix_out (B) int64 HOST: the partner permutation — groups of equal label in order of first
 *   appearance, each permuted by a FRESH Random(step).sample (bit-identical to CPython).
 * Does: permutation, boundary validation, packing, ONE hipMemcpyAsync, the kernel launch.
 * Returns 0; -1 frames not monotone / negative, -2 cycle end beyond T, -3 partner out of range
 * (nothing enqueued); or a hipError_t.
 */
int pcgmix_splice_same_label_f32(const float* x, float* y, const int64_t* labels,
                                 const int64_t* frames, uint64_t step, float lam,
                                 const double* knots, const double* spline_op, int n_knots,
                                 void* staging, void* dev_idx, int64_t* mix_out, int B, int C,
                                 int T, pcgmix_stream_t stream);
/* Same, with the labels still on the device as the reference passes them (one-hot int64
 * (B, num_classes), contiguous): copies them into `ohe_pinned` (PINNED host, B*num_classes
 * int64), synchronises `stream` (the one host sync the reference's signature forces,
 * augmentations.py:501), takes the first-maximum argmax and continues as above. */
int pcgmix_splice_same_label_ohe_f32(const float* x, float* y, const int64_t* target_ohe_dev,
                                     int num_classes, int64_t* ohe_pinned, const int64_t* frames,
                                     uint64_t step, float lam, const double* knots,
                                     const double* spline_op, int n_knots, void* staging,
                                     void* dev_idx, int64_t* mix_out, int B, int C, int T,
                                     pcgmix_stream_t stream);
long long pcgmix_splice_staging_bytes(int B, int C, int n_knots);

/* ------------------------------------------------------------------------------------------
 * Per-device step context + the whole fired step of a plain PCGmix method in ONE call.
 *                                                                            [host + device]
 * What the reference's augment() does between the probability gate and the returned tensor for
 * durratiomixup / durmixmagwarp with same-label partners (augmentations.py:874-928, 941-977):
 * label recovery from the one-hot matrix (:501), the partner permutation (:500-514), boundary
 * packing, the per-sample splice loop (:909-917, 969-977) and magnitude_warp (:674-683).
 *
 * The context owns what the step needs besides the batch: a ring of pinned staging buffers with
 * device twins (a slot is reused only after the kernel that read it has finished), host-mapped
 * memory for the label read-back, the spline operators per (T, n_knots).  It is created on
 * `device`; calls may come with any device current.  One context per device and host thread.
 *
 * pcgmix_augment_plain_f32:
 *   x, y            device (B, C, T); y must not alias x
 *   target_ohe_dev  device int64 (B, num_classes) one-hot as the reference passes it, or NULL when
 *   labels_host     host int64 (B) class labels are given instead (no read-back, no host wait)
 *   frames          HOST int64 (B, 5), as the reference's loader yields them
 *   lam             float32(lambda) drawn by the caller from numpy's global stream (:659-666)
 *   knots           HOST float64 (B, n_knots, C) as numpy.random.normal drew them (:677), or NULL
 *   mix_out         HOST int64 (B) out: the partner permutation (bit-identical to CPython's)
 * Order of work: the label arg-max kernel is enqueued FIRST (it writes int32 labels into
 * host-mapped memory and releases a flag word); boundaries are validated and packed, knots staged
 * and MT19937 seeded while the GPU gets there; then the host spins on the flag (the one wait
 * the reference's signature forces; falls back to hipStreamSynchronize after 2 ms), draws the
 * permutation, issues ONE hipMemcpyAsync and the fused kernel.  Must not be called on a
 * capturing stream when target_ohe_dev is used (hipErrorStreamCaptureUnsupported).
 * Returns 0; -1 frames not monotone / negative, -2 cycle end beyond T (splice not enqueued); or
 * a hipError_t.
 *
 * pcgmix_ctx_gate: random.Random(step).uniform(0, 1) (augmentations.py:869-870, 936-937); the
 * seeded generator is kept, so the step call for the same `step` does not seed a second time.
 */
/* Host-to-device copy as ONE kernel launch: src_pinned = pinned host memory (hipHostMalloc, torch
 * pin_memory: device-readable at the same address), dst_dev = device memory, both 16-byte aligned;
 * nbytes is rounded up to whole 16-byte words, which both buffers must hold.  For per-step index
 * data above 16 KB (e.g. the magnitude-warp knots, augmentations.py:677: 49 KB at bs 256):
 * hipMemcpyAsync switches from a blit kernel to the SDMA engine there, which on MI355X stalls the
 * stream for ~25 us per copy.  The step context does this internally.          */
int pcgmix_fetch_h2d(const void* src_pinned, void* dst_dev, size_t nbytes, pcgmix_stream_t stream);

typedef struct pcgmix_ctx pcgmix_ctx;
int pcgmix_ctx_create(int device, pcgmix_ctx** out);
void pcgmix_ctx_destroy(pcgmix_ctx* ctx);
double pcgmix_ctx_gate(pcgmix_ctx* ctx, uint64_t step);
/* Up to 1 MiB the caller wants on the device together with the NEXT pcgmix_augment_plain_f32 step
 * of this context (the float targets the loss reads — train_model.py:541-549 moves them with
 * their own .to(device) —, optimiser scalars, a dropout key): the bytes are copied now, appended
 * to that step's index block (same pinned slot, same single H2D copy) and written to dst_dev
 * (DEVICE, 16-byte aligned, >= bytes rounded up to 16) by the splice kernel, i.e. they are in
 * place for everything enqueued behind the step on its stream.  One shot; bytes == 0 withdraws. */
int pcgmix_ctx_set_payload(pcgmix_ctx* ctx, const void* host, size_t bytes, void* dst_dev);
/* Sends a pending payload on its own — the step it was meant for ran no plain splice (probability
 * gate): one H2D copy from the context's pinned ring straight to dst_dev on `stream`.  No-op
 * when nothing is pending. */
int pcgmix_ctx_flush_payload(pcgmix_ctx* ctx, pcgmix_stream_t stream);
/* The saliency-guided step ('(saloptenv)…' / '(saloptsum)…' with same-label partners,
 * augmentations.py:874-928) as two calls around the caller's saliency pass:
 *   begin   label arg-max kernel (labels -> host-mapped memory; float one-hot -> seed_out, DEVICE
 *           (B, num_classes), the saliency model's gradient seed, may be NULL); `frames` (HOST
 *           int64 (B,5)) validated, packed to int32 and copied to frames_dst_dev (DEVICE, B*5
 *           int32: what saliency post-processing, search and splice read).  target_ohe_dev may be
 *           NULL when the labels will be handed over to finish.
 *   ...     the caller enqueues the frozen model's forward + input gradient + post-processing
 *           (get_saliency_maps, saliency.py:20-91) and draws lambda and the warp knots
 *   finish  labels picked up (labels_host NULL) or taken from labels_host; partners drawn as
 *           get_same_label_mix_indices does; [partners | knots] in one H2D copy; displacement
 *           search + fused splice(+warp) as in pcgmix_salopt_mix_warp_f32; mix_out (HOST int64
 *           (B)) = the partner indices.
 * Returns 0, -1 / -2 for malformed boundaries (begin; nothing is copied), or a hipError_t.       */
int pcgmix_ctx_salopt_begin(pcgmix_ctx* ctx, const int64_t* target_ohe_dev, int num_classes,
                            float* seed_out, const int64_t* frames, int32_t* frames_dst_dev, int B,
                            int T, pcgmix_stream_t stream);
/* begin for a caller that holds the class labels on the HOST (a training loop with the loader's
 * CPU `target`, train_model.py:498-501): no read-back.  labels_host HOST int64 (B), values in
 * [0, num_classes).  Up to 256 samples (T <= 32767, num_classes <= 256) ONE launch carries the
 * labels, the boundaries and a pending pcgmix_ctx_set_payload of up to 320 bytes in its
 * arguments and writes seed_out, frames_dst_dev and the payload's destination; larger batches
 * stage [frames | labels] through a pinned slot (the payload then stays pending for
 * pcgmix_ctx_flush_payload).  finish is called with the same labels_host.  Returns as begin.
 * All step-context entry points return hipErrorStreamCaptureUnsupported on a capturing stream. */
int pcgmix_ctx_salopt_begin_labels(pcgmix_ctx* ctx, const int64_t* labels_host, int num_classes,
                                   float* seed_out, const int64_t* frames,
                                   int32_t* frames_dst_dev, int B, int T, pcgmix_stream_t stream);
int pcgmix_ctx_salopt_finish(pcgmix_ctx* ctx, const float* x, float* y, const float* sal,
                             const int32_t* frames_dev, const int64_t* labels_host, uint64_t step,
                             float lam, int mode, const double* knots, int n_knots, int64_t* mix_out,
                             int B, int C, int T, pcgmix_stream_t stream);
/* The label read-back alone, in two halves, for steps that have GPU work to enqueue in between
 * (the saliency-guided step, augmentations.py:881-907): begin = the arg-max kernel on `stream`;
 * wait = the spin on the flag word, then int64 class labels in labels_out (HOST, B).
 * seed_out (DEVICE, float (B, num_classes), may be NULL): the same kernel also writes the one-hot
 * of the labels as floats — d score[label] / d logits, the seed of the saliency model's input
 * gradient (saliency.py:52-61) — sparing the max / zeros / scatter launches that build it. */
int pcgmix_ctx_labels_begin(pcgmix_ctx* ctx, const int64_t* target_ohe_dev, int num_classes, int B,
                            float* seed_out, pcgmix_stream_t stream);
int pcgmix_ctx_labels_wait(pcgmix_ctx* ctx, int64_t* labels_out, int B, pcgmix_stream_t stream);
int pcgmix_augment_plain_f32(pcgmix_ctx* ctx, const float* x, float* y,
                             const int64_t* target_ohe_dev, int num_classes,
                             const int64_t* labels_host, const int64_t* frames, uint64_t step,
                             float lam, const double* knots, int n_knots, int64_t* mix_out,
                             int B, int C, int T, pcgmix_stream_t stream);

/* The armed plain step.  With target_ohe_dev given (the reference's signature: labels on the device,
 * augmentations.py:501), no warp, B <= 256, T <= 32767, T % 4 == 0 and 16-byte aligned x, y,
 * pcgmix_augment_plain_f32 enqueues ONE kernel at the start of the call: its first block does the
 * label arg-max, its other blocks wait for the index records the host writes into host-mapped
 * memory once it has drawn the partners (csrc/pcgmix_kernels.h, ArmedArgs).  Same output as the
 * two-launch path (PCGMIX_NO_ARMED=1), bit for bit.  The waiting blocks give up `timeout_ticks`
 * (100 MHz; default 1 s) after the kernel started; the call notices (records written later than
 * 0.4 s after the launch: stream synchronisation, abort word) and launches the splice again.
 * stats: out3 = steps armed | of them checked after a stream synchronisation | of them relaunched.
 * debug (tests): the relays' timeout and a host stall in front of the record write; 0, 0 = defaults. */
int pcgmix_ctx_armed_stats(pcgmix_ctx* ctx, long long* out3);
/* The armed plain step (no warp) in two calls, for a binding with host work of its own between the
 * launch and the moment lambda is known — PKG/augmentations.py draws lambda from numpy's global stream
 * there (augmentations.py:661-663), 2.7 us that otherwise stand in front of the launch.
 * begin: validates, and if the step is eligible (conditions above) enqueues the armed kernel and returns
 * 0; returns PCGMIX_NOT_ARMED with nothing enqueued when it is not (call pcgmix_augment_plain_f32), a
 * hipError_t on failure.  A pending pcgmix_ctx_set_payload travels with the launch.
 * finish: boundaries (validated: -1 / -2 as pcgmix_augment_plain_f32, the waiting kernel is released),
 * label pick-up, partner draw into mix_out (HOST int64 (B)), records + lambda.  hipErrorNotReady without
 * a begin.  A begin that is never finished is released by the next begin (or by the blocks' timeout). */
#define PCGMIX_NOT_ARMED (-3)
int pcgmix_augment_plain_begin(pcgmix_ctx* ctx, const float* x, float* y, const int64_t* target_ohe_dev,
                               int num_classes, int B, int C, int T, pcgmix_stream_t stream);
int pcgmix_augment_plain_finish(pcgmix_ctx* ctx, const int64_t* frames, uint64_t step, float lam,
                                int64_t* mix_out);
int pcgmix_ctx_armed_debug(pcgmix_ctx* ctx, unsigned long long timeout_ticks, int stall_ms);

/* Diagnostic: mean host nanoseconds per pcgmix_augment_plain_f32 call since the last query, by
 * phase: label kernel launch | slot reserve | pack + seed | label wait | grouping + permutation |
 * H2D enqueue | kernel launch | event record.  Returns the number of calls averaged; resets. */
long long pcgmix_ctx_phase_times(pcgmix_ctx* ctx, double* out8);

/* Which instantiation of the splice kernel pcgmix_mix_warp_f32 launches for this problem:
 * vec = 4 (16-byte lanes; needs T % 4 == 0 and 16-byte aligned x, y) or 1, unroll = quads per
 * lane (1, 2, 4).  For reporting: the kernel's name is mix_warp_kernel<vec, warp, unroll>. */
int pcgmix_mix_variant(int B, int C, int T, int warp, int aligned16, int* vec, int* unroll);
/* Name of the kernel instantiation pcgmix_mix_warp_f32 launches for this problem, as rocprofv3
 * lists it (e.g. "pcgmix::mix_warp_tq_kernel<2, 1>"), written NUL-terminated into buf.  n_knots = 0:
 * no warp; zero_rect != 0: the call carries 2D mask rectangles.  Asks the same selection routine
 * the launcher uses. */
int pcgmix_mix_kernel_name(int B, int C, int T, int n_knots, int zero_rect, int aligned16, char* buf,
                           int buf_len);

/* ------------------------------------------------------------------------------------------
 * BatchNorm (training mode) + ReLU + MaxPool of a ResNet9 block, channels innermost.  [device]
 *
 * Replaces nn.BatchNorm1d/2d -> nn.ReLU -> nn.MaxPool1d/2d of conv_block (models.py:468-473,
 * models2d.py:13-19) and their autograd.  y is the convolution output stored NHWC: (B, H, W, C)
 * row-major (H = 1 for the 1D network), C % 4 == 0 and 256 % (C / 4) == 0 (64 ... 1024).
 *   forward : batch mean / biased variance per channel -> z (B, H/ph, W/pw, C) =
 *             maxpool_{ph x pw}(relu(gamma * (y - mean) / sqrt(var + eps) + beta)) [+ skip];
 *             ph = pw = 1: no pooling.  skip (shape of z, or NULL): the residual connection of
 *             res1 / res2 (models.py:577, 581: `out = self.res1(out) + out`) added in the same
 *             pass; its gradient is dz itself, so the backward below is unchanged.  mean, invstd (C each) are outputs kept for backward; running_mean /
 *             running_var (may be NULL) are updated in place with `momentum` (unbiased variance).
 *   backward: dz (shape of z) -> dx (shape of y), dgamma, dbeta (C each).  The ReLU mask and the
 *             pooling arg-max (first maximum) are recomputed from y.
 * workspace: pcgmix_bnrp_workspace_floats(B, H, W, C) floats.  All pointers 16-byte aligned.
 * Each direction reads y twice and writes its output once; reductions are fixed-order.
 */
long long pcgmix_bnrp_workspace_floats(int B, int H, int W, int C);
/* mean_shift (C, may be NULL): a per-channel constant left out of y — the convolution's bias, which
 * the normalisation cancels — added to the batch mean in the running-mean update only;
 * batches_tracked (may be NULL): nn.BatchNorm's num_batches_tracked, incremented by one;
 * dzero (C, may be NULL): receives zeros, the exact gradient of such a constant (so that the
 * optimiser still sees the parameter and applies weight decay, as in the reference).            */
int pcgmix_bnrp_fwd_f32(const float* y, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float momentum, float eps, const float* mean_shift,
                        long long* batches_tracked, const float* skip, float* z, float* mean,
                        float* invstd, float* workspace, int B, int H, int W, int C, int ph, int pw,
                        pcgmix_stream_t stream);
int pcgmix_bnrp_bwd_f32(const float* y, const float* dz, const float* gamma, const float* beta,
                        const float* mean, const float* invstd, float* dx, float* dgamma,
                        float* dbeta, float* dzero, float* workspace, int B, int H, int W, int C,
                        int ph, int pw, pcgmix_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PCGMIX_HIP_H */
