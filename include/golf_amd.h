/*
 * golf_amd.h — C ABI of libgolf_hip.so: the MI355X (gfx950) kernels behind GOLF's time-varying
 * LPC synthesis filter and glottal-flow source path.
 *
 * The reference (iamycy/golf) has NO FFI boundary of its own: the arithmetic on this path lives in
 * un-vendored Python/C++/numba packages that its nn.Modules call.  Each entry point below names the
 * reference call site (file:line under /root/reference) whose computation it replaces; the Python
 * binding a maintainer would add is golf_amd/_lib.py (ctypes) — see INTEGRATION.md.
 *
 * Conventions (all entry points)
 *   - plain pointers + sizes only; every pointer is a caller-owned DEVICE buffer (hipMalloc'd or a
 *     torch tensor's data_ptr()), fp32 row-major contiguous unless a stride is given;
 *   - the library never allocates or frees: scratch is passed in as (ws, ws_bytes), sized by the
 *     matching *_workspace_bytes();  ws must be 256-byte aligned;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*), no internal sync,
 *     no global mutable state => re-entrant from several host threads / streams;
 *   - return 0 = ok; <0 = GOLF_E* bad-argument code (text via golf_last_error(), thread-local);
 *     >0 = hipError_t from a launch.
 */
#ifndef GOLF_AMD_H
#define GOLF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOLF_ABI_VERSION 6

enum {
    GOLF_OK = 0,
    GOLF_EINVAL = -1,   /* bad size / null pointer                        */
    GOLF_EWORKSPACE = -2, /* ws too small or misaligned                     */
    GOLF_EUNSUPPORTED = -3 /* shape outside what the kernels cover (message) */
};

int golf_abi_version(void);
const char* golf_last_error(void);
/* "gfx950" — the only code object in the library. */
const char* golf_target_arch(void);

/* ---------------------------------------------------------------------------------------------
 * a-1 (+a-3 fused): sample-wise LTV all-pole filter — GOLF-ss end filter.
 * Replaces LTVMinimumPhaseFilterPrecise.forward, models/filters.py:99-113:
 *     ex*gain (AudioTensor broadcast -> linear upsampling, models/utils.py:538-544)
 *     a.reduce_hop_length() -> (B,T,M) materialised coefficient tensor (never built here)
 *     torchlpc.sample_wise_lpc(ex, a)  (models/filters.py:112)
 *
 *   y[b,t] = ex[b,t]*G[b,t] - sum_{i<M} A[b,t,i]*y[b,t-1-i],  y[<0] = 0,  t in [0,T)
 *   G = up(gain), A = up(a): up(z)[t] = z[f]+(t-f*hop)*(z[f+1]-z[f])/hop, f = min(t/hop, F-2)
 *
 *   ex   (B, >=T) row stride ex_stride      gain (B,F)      a (B,F,M)
 *   y    (B, T)   row stride y_stride       requires 1 <= T <= (F-1)*hop+1, 1 <= M <= 64
 *   ws   scratch of golf_ltv_allpole_workspace_bytes(); it also carries the per-chunk transition
 *        matrices the backward pass reuses — keep it alive (unmodified) until the backward ran.
 * ------------------------------------------------------------------------------------------- */
size_t golf_ltv_allpole_workspace_bytes(int B, int T, int F, int M, int hop);
/* The same for a forced algorithm (GOLF_SS_SERIAL / GOLF_SS_CHUNKED in `flags`; 0 = default selection). */
size_t golf_ltv_allpole_workspace_bytes_ex(int B, int T, int F, int M, int hop, int flags);

/* The per-chunk transition matrices depend only on the coefficients `a`, not on the excitation, and are the
 * most expensive phase.  golf_ltv_allpole_transitions_f32 computes them into `ws` on its own, so a caller that
 * knows `a` before `ex` exists can start it early (optionally on a second HIP stream) or reuse it for several
 * excitations, and then pass GOLF_SS_HAVE_TRANSITIONS to the forward.
 *   The transitions call also prepares what the forward's boundary scan derives from the matrices alone (the group
 *   composites of the two-level scan): give BOTH calls the same algorithm flags (GOLF_SS_CHUNKED / _FLAT_SCAN / _FAST_...),
 *   and size `ws` with golf_ltv_allpole_workspace_bytes_ex for those flags.
 *   flags  GOLF_SS_HAVE_TRANSITIONS  `ws` already holds the transitions for (a,B,T,F,M,hop)
 *          GOLF_SS_FAST_TRANSITIONS  inference mode: transition matrices from fp32 instead of fp64 trajectories
 *                (about 4x cheaper) and the forward runs one refinement sweep over the chunk boundary states
 *                (re-run the chunks from the scanned states, rescan with the observed end-state defects), which
 *                makes the result second order in the matrix error: same accuracy as a sequential fp32 recursion.
 *                (ABI 2 required the fp64-derived matrices for the backward; since ABI 3 the backward runs its own refinement
 *                sweep and accepts either: see GOLF_SS_TRAINING.)
 *   side_stream  optional second hipStream_t (may be NULL): without HAVE_TRANSITIONS the forward forks the
 *                transition kernel onto it (event fork/join) so it overlaps the excitation-dependent phase;
 *                with HAVE_TRANSITIONS the forward joins it (event wait) right before the boundary scan. */
#define GOLF_SS_HAVE_TRANSITIONS 1
#define GOLF_SS_FAST_TRANSITIONS 2
/*          GOLF_SS_SPLIT_P1  (diagnostic) launch the fp32 transition kernel and the zero-state pass separately instead
 *                of as one horizontally fused kernel (the default with FAST and without HAVE / side_stream). */
#define GOLF_SS_SPLIT_P1 4
/*          GOLF_SS_SERIAL / GOLF_SS_CHUNKED  force one of the two algorithms (default: by batch size).
 *                CHUNKED: time is cut into chunks whose transition matrices are scanned (B*T/L*(M+2) lanes of work,
 *                (M+2)-fold arithmetic: what makes B = 32 fast).  SERIAL: one quad of lanes per utterance runs the
 *                recursion from t = 0 to T, B/16 waves, no redundant arithmetic and no transition matrices: the
 *                large-batch kernel (default from B >= 2048, where the two meet on MI355X).  The backward must be given the flag the forward ran with. */
#define GOLF_SS_SERIAL 8
#define GOLF_SS_CHUNKED 16
/*          GOLF_SS_FLAT_SCAN  chunk-boundary states by the flat scan (one wave per utterance, NP dependent matvecs)
 *                instead of the two-level scan that long utterances with M <= 24 take by default while the batch is
 *                small (utterances x groups of 16 chunks <= 2 x the CU count: B <= 39 at 2 s; group composites as
 *                f64 MFMA product chains rounded once to fp32 + per-group scans spread over the chip + start states
 *                derived in the chunk kernels): an A/B switch; the two give the same states up to fp32 rounding. */
#define GOLF_SS_FLAT_SCAN 32
/*          GOLF_SS_TRAINING  golf_ltv_allpole_bwd_f32 will follow on this workspace: the forward also keeps what the backward's
 *                two-level adjoint scan reads (the transition matrices in the adjoint's orientation, the group composites
 *                transposed).  Implied without GOLF_SS_FAST_TRANSITIONS.  With FAST | TRAINING the training step runs on the
 *                fp32 transition matrices of the inference path (hot chunks recomputed from fp64 trajectories, one refinement
 *                sweep in the forward AND in the backward): same accuracy class, ~28 us less per B = 32 step (ABI 3). */
#define GOLF_SS_TRAINING 64
/*          GOLF_SS_MAPS_ONLY  (ABI 4, with GOLF_SS_FAST_TRANSITIONS) golf_ltv_allpole_transitions_f32 computes the transition
 *                matrices and nothing else; the forward, given HAVE_TRANSITIONS | MAPS_ONLY, runs what is still missing (the
 *                fix-up of ill-conditioned matrices, the group composites) in the launch that holds its zero-state pass.
 *                The matrices need only `a`: a caller that has the coefficient tracks before the excitation (the decoder: they
 *                come from the encoder, the excitation from the oscillator) issues this call beside the source's launches
 *                (golf_amd.functional.ltv_allpole_prepare(maps_only=True)). */
#define GOLF_SS_MAPS_ONLY 128
/*          GOLF_SS_THROUGHPUT  (ABI 4) the caller keeps SEVERAL batches in flight on the device (a serving loop on a few HIP
 *                streams): prefer the launch structure that costs the least chip time over the one that finishes a lone
 *                batch soonest.  Today: the zero-state pass runs inside the pre-pass launch instead of inside the transition
 *                kernel's, and the two chunk passes are two thin launches instead of the one launch a lone batch gets (whose
 *                waves live through both sweeps, waiting for each other in between) -- one batch alone 123 -> 134 us, four in
 *                flight 75 -> 68.5 us/step, MI355X, B = 32 x 2 s (round 5).  Results are bit-identical either way: the flag changes launches, never
 *                the algorithm.  (The Python host additionally switches an inference forward of >= 512 utterances to GOLF_SS_SERIAL
 *                while it sets this flag -- golf_amd.functional.SS_THROUGHPUT_SERIAL_MIN -- and that IS another algorithm: same
 *                accuracy class, different bits.)  The library cannot see how many batches its
 *                caller keeps in flight, hence a flag (cf. GOLF_SS_SERIAL). */
#define GOLF_SS_THROUGHPUT 256
/*          GOLF_SS_ZERO_TAIL  (ABI 5, golf_ltv_allpole_bwd_f32 only) the excitation rows were longer than the output (the
 *                oscillator renders Tx = 48 000 samples, the filter consumes T = (F-1)*hop + 1 = 47 761): the backward also
 *                writes the zeros that are the gradient of the unused tail, g_ex[b][T .. g_ex_stride), so that the caller
 *                needs neither a full-size memset nor a strided fill launch in front of every backward (each ~5 us of a
 *                ~260 us B = 32 training step).  g_ex must be dense up to its row stride. */
#define GOLF_SS_ZERO_TAIL 512

int golf_ltv_allpole_transitions_f32(const float* a, int B, int T, int F, int M, int hop,
                                     void* ws, size_t ws_bytes, int flags, void* stream);

int golf_ltv_allpole_fwd_f32(const float* ex, int64_t ex_stride, const float* gain, const float* a,
                             float* y, int64_t y_stride, int B, int T, int F, int M, int hop,
                             void* ws, size_t ws_bytes, int flags, void* side_stream, void* stream);

/* Conditioning and health status of the forward that last used `ws` (ABI 3).  The reference's sequential fp32
 * recursion degrades gracefully when an interpolated filter comes close to instability (models/filters.py:99-113 ->
 * torchlpc.sample_wise_lpc); the time-chunked algorithm keeps that behaviour by recomputing the
 * transition matrices of such chunks from fp64 trajectories and, for the worst utterances, scanning their boundary
 * states in fp64 (csrc/lpc_ss.hip, "conditioning tiers").  This call reports how often that happened and whether the
 * output is finite -- asynchronously, on `stream`, into 4 caller-owned DEVICE words the host reads when it likes:
 *   out[0]  utterances with at least one chunk whose matrix was recomputed from fp64 trajectories
 *   out[1]  utterances whose boundary states came from the fp64 scan (tier 3)
 *   out[2]  bit 0: a non-finite sample was written to y (an unstable filter is surfaced, not hidden);
 *           bit 1: a wait inside the pre-pass launch for the recomputed matrices ran into its bound (never observed; the
 *           result is then the fp32 matrices' -- less accurate, not undefined);
 *           bit 2: golf_ltv_allpole_bwd_f32 ran on this workspace with another boundary scan than the forward that filled
 *           it (GOLF_SS_FLAT_SCAN / _CHUNKED bits that differ between the two calls): its gradients are void.  Call this
 *           function after the backward to see it.
 *   out[3]  the largest |entry| over all transition matrices of the batch, as the bits of an fp32
 * (B,T,F,M,hop,flags) as given to the forward.  The serial / generic algorithms form no matrices: all four words 0. */
int golf_ltv_allpole_status_u32(const void* ws, size_t ws_bytes, int B, int T, int F, int M, int hop, int flags,
                                uint32_t* out, void* stream);

/* Custom backward of the above (what torchlpc's autograd.Function + autograd through
 * F.interpolate compute in the reference; closed form in SURVEY.md App. A-2):
 *     g[t]      = gy[t] - sum_i A[t+1+i,i]*g[t+1+i]      (reverse-time recursion)
 *     g_ex[t]   = g[t]*G[t]
 *     g_gain[f] = up^T(g*ex)[f]         g_a[f,i] = up^T(-g[t]*y[t-1-i])[f,i]
 *   gy,y (B,T) with strides; ws = the forward's workspace (same B,T,F,M,hop); flags = the forward's
 *   GOLF_SS_SERIAL / GOLF_SS_CHUNKED bits (ABI 2);
 *   g_ex (B,T) stride g_ex_stride, g_gain (B,F), g_a (B,F,M) are fully overwritten. */
int golf_ltv_allpole_bwd_f32(const float* gy, int64_t gy_stride, const float* y, int64_t y_stride,
                             const float* ex, int64_t ex_stride, const float* gain, const float* a,
                             float* g_ex, int64_t g_ex_stride, float* g_gain, float* g_a,
                             int B, int T, int F, int M, int hop,
                             void* ws, size_t ws_bytes, int flags, void* stream);

/* a-5: inverse (analysis) filter e[t] = y[t] + sum_i A[t,i]*y[t-1-i].
 * Replaces LTVMinimumPhaseFilter.reverse -> fir_filt, models/filters.py:186-195, utils.py:433-441. */
int golf_ltv_inverse_f32(const float* y, int64_t y_stride, const float* a, float* e, int64_t e_stride,
                         int B, int T, int F, int M, int hop, void* stream);
/* Its backward (autograd through fir_filt's unfold + matmul in the reference; used when a decoder is trained with
 * `inverse_target`, ltng/vocoder.py:192-200):  g_y[t] = g_e[t] + sum_i A[t+1+i,i] g_e[t+1+i],
 * g_a[f,i] = up^T(g_e[t] * y[t-1-i]).  Either output may be NULL. */
int golf_ltv_inverse_bwd_f32(const float* g_e, int64_t g_e_stride, const float* y, int64_t y_stride, const float* a,
                             float* g_y, int64_t g_y_stride, float* g_a, int B, int T, int F, int M, int hop,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * a-4: frame-wise LTI all-pole + windowed overlap-add — GOLF-ff end filter.
 * Replaces LTVMinimumPhaseFilter.forward, models/filters.py:131-184 (pad, unfold, lpc_synthesis ->
 * torchaudio.functional.lfilter models/lpc.py:11-16, diagonal conv_transpose1d OLA, normalise).
 *   x = ex*up(gain) (length Tx = min(T_ex, (F-1)*hop+1)), zero-padded by W/2 each side;
 *   frame f = x_pad[f*hop .. f*hop+W), nfr = (Tx + 2*(W/2) - W)/hop + 1 <= F frames,
 *   filt_f = LTI all-pole a[b,f,:] from zero state;
 *   y[n] = sum_f window[k]*filt_f[k] / sum_f window[k],  n = f*hop - W/2 + k,  n in [0, Ty),
 *   Ty = (nfr-1)*hop + W - 2*(W/2).
 *   window (W) fp32; requires W >= 2*hop.  `centred==0` handling (drop hop/2, reflect pad) is done
 *   by the host wrapper.  ws: golf_lti_frames_workspace_bytes(). */
size_t golf_lti_frames_workspace_bytes(int B, int Tx, int F, int M, int hop, int W);

int golf_lti_frames_ola_fwd_f32(const float* ex, int64_t ex_stride, const float* gain, const float* a,
                                const float* window, float* y, int64_t y_stride,
                                int B, int Tx, int F, int M, int hop, int W, int Ty,
                                void* ws, size_t ws_bytes, void* stream);

/* Custom backward of the above (what autograd computes in the reference through conv_transpose1d, lfilter, unfold,
 * the zero pad and the gain product; closed form in oracle/golf_oracle.py::lti_frames_ola_backward, pinned by
 * tests/golden/g15):  g_q = gy/norm;  u_f = the frame's all-pole recursion run backwards in time on window*g_q;
 *     g_a[f,i] = -sum_k u_f[k]*y_f[k-1-i];   g_x = overlap-add of the u_f;   g_ex = g_x*G;   g_gain = up^T(g_x*ex).
 *   ws_fwd = the forward's workspace, unmodified (it holds the filtered frames y_f);
 *   ws     = scratch of golf_lti_frames_bwd_workspace_bytes();
 *   g_ex (B, g_ex_len >= Tx) is fully written (zeros past Tx and past sample (F-1)*hop: ABI 5); g_gain (B,F) and
 *   g_a (B,F,M) are fully overwritten.  Requires W % ring width == 0 (the fast path of the forward). */
size_t golf_lti_frames_bwd_workspace_bytes(int B, int Tx, int F, int M, int hop, int W);
int golf_lti_frames_ola_bwd_f32(const float* gy, int64_t gy_stride, const float* ex, int64_t ex_stride,
                                const float* gain, const float* a, const float* window, float* g_ex,
                                int64_t g_ex_stride, int g_ex_len, float* g_gain, float* g_a, int B, int Tx, int F,
                                int M, int hop, int W, int Ty, const void* ws_fwd, void* ws, size_t ws_bytes,
                                void* stream);

/* a-6: the same frame-wise synthesis with the all-pole filter given as a CASCADE of K second-order sections.
 * Replaces BatchSecondOrderLPCSynth.forward, models/lpc.py:94-131 (pad, unfold, K successive torchaudio lfilter calls
 * with a = biquads[...,k,:] and b = [1,0,0], windowed overlap-add, normalisation).
 *   biquads (B,F,K,3) = (a0,a1,a2) per section, K <= 16;  frames of W samples every hop of the signal zero-padded by
 *   `pad` on both sides (the reference class uses (W-hop)/2; LTVMinimumPhaseFilter's convention is W/2);
 *   gain_mode 1: frame f is scaled by gain[b,f] (the reference class); 0: ex is multiplied by up(gain) first.
 *   nfr = (Tx + 2*pad - W)/hop + 1 <= F,  Ty = (nfr-1)*hop + W - 2*pad;  ws: golf_lti_frames_workspace_bytes(). */
int golf_biquad_frames_ola_fwd_f32(const float* ex, int64_t ex_stride, const float* gain, const float* biquads,
                                   const float* window, float* y, int64_t y_stride, int B, int Tx, int F, int K,
                                   int hop, int W, int pad, int gain_mode, int Ty, void* ws, size_t ws_bytes,
                                   void* stream);

/* Backward of the cascade (the reference's is autograd through its K lfilter calls, models/lpc.py:115-118; closed form in
 * oracle/golf_oracle.py::biquad_frames_ola_backward, pinned by the reference's own gradients in tests/golden/g26):
 *   g_q = gy / norm (the caller divides: norm = overlap-add of the window);  per frame, u_K = window * g_q and
 *   u_{k-1} = section k run BACKWARDS in time on u_k;  g_biquads[b,f,k,i] = -sum_n u_{k-1}[n] * y_k[n-i] (y_k = the section's
 *   output, recomputed);  the frames' u_0 are overlap-added:
 *   gain_mode 1: g_ex = that sum (each frame scaled by its gain), g_gain_frames[b,f] = sum_n u_0[n] * x_f[n];
 *   gain_mode 0: g_ex = sum * up(gain), and gx_ex (B,Tx) = sum * ex for the caller to fold onto the gain frames (up^T).
 *   g_ex (B,Tx), g_biquads (B,F,K,3) are fully overwritten; ws: golf_biquad_frames_bwd_workspace_bytes(). */
size_t golf_biquad_frames_bwd_workspace_bytes(int B, int Tx, int F, int K, int hop, int W, int pad);
int golf_biquad_frames_ola_bwd_f32(const float* gq, int64_t gq_stride, const float* ex, int64_t ex_stride,
                                   const float* gain, const float* biquads, const float* window, float* g_ex,
                                   int64_t g_ex_stride, float* g_gain_frames, float* g_biquads, float* gx_ex,
                                   int B, int Tx, int F, int K, int hop, int W, int pad, int gain_mode, int Ty,
                                   void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a-2: control transform of the LPC filters, logits -> direct-form coefficients.
 * Replaces rc2lpc(tanh(logits) * max_abs_value), models/filters.py:91-97 + models/utils.py:581-593 (a Python loop of
 * M-1 Levinson step-up iterations A_n = [A_{n-1}, 0] + k_n * flip([A_{n-1}, 0]), ~100 tiny kernels) by one kernel.
 *   logits (N, M) with N = B*F frames, contiguous;  a (N, M) = a_1..a_M;  M <= 64;
 *   apply_tanh != 0: k = tanh(logits) * max_abs;  0: k = logits (reflection coefficients given directly).
 * Backward: g_logits (N, M) from g_a (N, M) (the adjoint of the step-up, through the tanh when apply_tanh). */
int golf_rc2lpc_fwd_f32(const float* logits, float* a, int64_t N, int M, float max_abs, int apply_tanh,
                        void* stream);
int golf_rc2lpc_bwd_f32(const float* logits, const float* g_a, float* g_logits, int64_t N, int M, float max_abs,
                        int apply_tanh, void* stream);

/* The biquad parameterisations of the ISMIR'23 configs, logits (N, 2K) -> K second-order sections -> their product in
 * direct form a (N, 2K).  Replaces get_logits2biquads(rep)(logits) + biquads2lpc, models/utils.py:487-525 and :444-484
 * (models/filters.py:71-81), 112 kernels in PyTorch ops.  rep: 0 "coef", 1 "conj", 2 "real"; 2K <= 64. */
int golf_sos2lpc_fwd_f32(const float* logits, float* a, int64_t N, int K, float max_abs_pole, int rep, void* stream);
int golf_sos2lpc_bwd_f32(const float* logits, const float* g_a, float* g_logits, int64_t N, int K, float max_abs_pole,
                         int rep, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a-8/a-9: indexed glottal-flow wavetable oscillator.
 * Replaces IndexedGlottalFlowTable.forward, models/synth.py:213-263 (table blend, phase/oversampling,
 * linear upsample, fp32 cumsum, %1, GlottalFlowTable.generate = F.grid_sample bilinear
 * models/synth.py:124-177, * rsqrt(phase), kazane.Decimate(oversampling)).
 *   phase (B,Tp) phase increment [cycles/sample] at hop phase_hop; wsel (B,Fw) in [0,1] at hop w_hop;
 *   table (n_tab, L).  Oversampled length N = (Tp-1)*phase_hop*os + 1 (N = Tp if phase_hop*os==1).
 *   pre (B,N) (optional, may be NULL) = the signal handed to the decimator;
 *   out (B,Tout): os==1 -> out = pre (Tout = N); os>1 -> strided FIR with `taps` (K odd),
 *   Tout = (N-1)/os + 1.
 *   The running phase is accumulated exactly (64-bit fixed point; the reference uses an fp32
 *   cumsum; parity is against the float64 oracle).
 *   addend (B, Tadd rows of stride addend_stride; optional, may be NULL; os > 1 only): out[b,o] += addend[b,o] for
 *   o < Tadd — the `harm_osc + noise_filter(noise)` of SourceFilterSynth.forward, models/sf.py:53-56, fused into the
 *   decimator's epilogue (one full-tensor round trip and one launch less per step). */
size_t golf_glottal_osc_workspace_bytes(int B, int Tp, int phase_hop, int Fw, int w_hop, int L, int os);

/* ABI 6: the decimation taps laid out for the matrix pipe, ONCE per tap set instead of once per step.  The fused oscillator
 * (os = 4, phase at hop 1, power-of-two table) multiplies the 4x oversampled signal by Toeplitz fragments of the taps
 * (v_mfma_f32_16x16x4_f32); they are a function of the taps alone -- kazane.Decimate's kernel is a constant buffer of the module,
 * models/synth.py:208-211 -- so a caller prepares them when the taps change and hands them to every forward / backward as
 * `tap_frags`.  NULL is always allowed: the call then lays them out itself (one small launch more).  A backward with
 * GOLF_OSC_WS_KEPT takes the same `tap_frags` its forward took.  _bytes returns 0 where the fused path does not apply. */
size_t golf_glottal_osc_tap_fragments_bytes(int K, int os);
int golf_glottal_osc_tap_fragments_f32(const float* taps, int K, int os, void* frags, size_t frags_bytes, void* stream);

/* equal_energy | GOLF_OSC_THROUGHPUT (ABI 6, forward only): the caller keeps SEVERAL batches in flight on the device (cf.
 * GOLF_SS_THROUGHPUT): below a device-filling batch the phase scan then runs as a small launch of its own in front of the
 * fused kernel instead of inside it -- measured cheaper by ~1.3 us per B = 32 step with four batches in flight, equal for a
 * lone batch, 3 % dearer at B = 16 384 (where the flag changes nothing).  Results are bit-identical either way. */
#define GOLF_OSC_THROUGHPUT 4
/* Round 6: ONE launch for the fused configuration.  The running phase is a single-pass scan inside the kernel (decoupled
 * look-back over tagged entries in `ws`; no totals launch, the phase is read once) -- the workspace may hold anything on entry
 * (no zero-fill is required, stale entries of earlier launches never validate) but must not be shared by launches in flight
 * at the same time. */
int golf_glottal_osc_fwd_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                             const float* wsel, int Fw, int w_hop,
                             const float* table, int n_tab, int L,
                             int os, int equal_energy, const float* taps, int K,
                             float* pre, float* out, int64_t out_stride, int B, int Tout,
                             void* ws, size_t ws_bytes, void* stream,
                             const float* addend, int64_t addend_stride, int Tadd, const void* tap_frags);

/* ABI 6 (round 6): the source and the end filter's transition maps in ONE launch -- SourceFilterSynth.forward's
 * `harm_osc(...) + noise` followed by `end_filter(...)`, models/sf.py:47-64, where the filter's chunk transition maps
 * (golf_ltv_allpole_transitions_f32) depend on the coefficients only and not on the source.
 *   == golf_glottal_osc_fwd_f32(phase .. tap_frags; pre = NULL)  then
 *      golf_ltv_allpole_transitions_f32(a, B, T, F, M, hop, ss_ws, ss_ws_bytes, ss_flags, stream)
 * bit for bit; as one grid (oscillator workgroups beside the transition-map waves: a lone B = 32 batch ~42 us instead of
 * 6 + 18 + 38) when ss_flags holds GOLF_SS_FAST_TRANSITIONS | GOLF_SS_MAPS_ONLY, the oscillator's fused configuration applies
 * and the filter runs on its 24-sample ring with 22 taps (lpc_order 19 .. 22, hop % 24 == 0); as the two calls otherwise.
 * The caller then runs golf_ltv_allpole_fwd_f32(out as ex, .., ss_ws, ss_flags | GOLF_SS_HAVE_TRANSITIONS).
 * osc_ws / ss_ws: the two calls' workspaces (golf_glottal_osc_workspace_bytes / golf_ltv_allpole_workspace_bytes_ex). */
int golf_source_transitions_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                                const float* wsel, int Fw, int w_hop,
                                const float* table, int n_tab, int L,
                                int os, int equal_energy, const float* taps, int K,
                                float* out, int64_t out_stride, int B, int Tout,
                                void* osc_ws, size_t osc_ws_bytes,
                                const float* addend, int64_t addend_stride, int Tadd, const void* tap_frags,
                                const float* a, int T, int F, int M, int hop,
                                void* ss_ws, size_t ss_ws_bytes, int ss_flags, void* stream);

/* Backward w.r.t. table_select_weight only (phase is data in GOLF training: train_with_true_f0,
 * cfg/ae/vctk.yaml:72):  g_wsel (B,Fw) overwritten.  ws = a workspace of the forward's size.
 * equal_energy | GOLF_OSC_WS_KEPT (ABI 5): the caller vouches that `ws` is exactly as golf_glottal_osc_fwd_f32 left it for
 * these same arguments with pre == NULL (an autograd node that saved it): the backward then reuses the forward's phase tile
 * totals and tap fragments instead of recomputing them (one launch of three less, ~6 us at B = 32).  Without the flag nothing
 * is assumed about the workspace's contents. */
#define GOLF_OSC_WS_KEPT 2
int golf_glottal_osc_bwd_wsel_f32(const float* g_out, int64_t g_out_stride,
                                  const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                                  const float* wsel, int Fw, int w_hop,
                                  const float* table, int n_tab, int L,
                                  int os, int equal_energy, const float* taps, int K,
                                  float* g_wsel, int B, int Tout,
                                  void* ws, size_t ws_bytes, void* stream, const void* tap_frags);

/* Generic wavetable lookup = GlottalFlowTable.generate, models/synth.py:124-177 (F.grid_sample bilinear over
 * (control frame, phase)), for arbitrary per-frame tables (B,K,L) at hop hop_t and a given wrapped phase (B,N) in [0,1):
 * what WeightedGlottalFlowTable (:266-294), WrappedPhaseDownsampledIndexedGlottalFlowTable (:343-375) and
 * IndexedGlottalFlowTable with a differentiable phase / phase_offset / trainable table (:59-70,:213-218) reduce to.
 *   out[b,n] = bilerp(T[b], row n/hop_t (frames beyond K-1 replicate the last), column wrapped*L (wraps to column 0))
 * Backward: g_wrapped (B,N) and/or g_tables (B,K,L) (either may be NULL; both are fully overwritten). */
int golf_wavetable_lookup_fwd_f32(const float* wrapped, int64_t wrapped_stride, const float* tables, int K, int L,
                                  int hop_t, float* out, int64_t out_stride, int B, int N, void* stream);
int golf_wavetable_lookup_bwd_f32(const float* g_out, int64_t g_out_stride, const float* wrapped, int64_t wrapped_stride,
                                  const float* tables, int K, int L, int hop_t, float* g_wrapped,
                                  int64_t g_wrapped_stride, float* g_tables, int B, int N, void* stream);

/* The running phase of those general table oscillators (ABI 3):
 *   wrapped[b,n] = frac( cumsum(up(phase / os))[n] + phase_offset[b,n] ),   n < N <= (Tp-1)*phase_hop*os + 1
 * Replaces F.interpolate + torch.cumsum (float32 in the reference) + % 1 of IndexedGlottalFlowTable.forward,
 * models/synth.py:239-255, with the exact 64-bit fixed-point prefix of the fused oscillator.  phase_offset (B,>=N) at the
 * fine rate, or NULL.  The gradient w.r.t. phase is the transposed upsampling of the reverse cumulative sum (host). */
size_t golf_phase_accumulate_workspace_bytes(int B, int Tp);
int golf_phase_accumulate_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop, int os,
                              const float* phase_offset, int64_t offset_stride, float* wrapped, int64_t wrapped_stride,
                              int B, int N, void* ws, size_t ws_bytes, void* stream);

/* The oscillator's decimator on its own (kazane.Decimate(os) stand-in, models/synth.py:208,262; taps are an input):
 *   out[b,o] = sum_k taps[k] * x[b, o*os + k - (K-1)/2], zero padded, Tout = (N-1)/os + 1, K odd, os in [2,64];
 * golf_decimate_fir_adj_f32 is its transpose (g_x (B,N) dense, fully overwritten). */
int golf_decimate_fir_f32(const float* x, int64_t x_stride, int N, const float* taps, int K, int os, float* out,
                          int64_t out_stride, int B, int Tout, void* stream);
int golf_decimate_fir_adj_f32(const float* g_out, int64_t g_out_stride, int Tout, const float* taps, int K, int os,
                            float* g_x, int N, int B, void* stream);

/* ---------------------------------------------------------------------------------------------
 * f-1: zero-phase FIR noise filter of every GOLF decoder.
 * Replaces LTVZeroPhaseFIRFilter.forward, models/filters.py:340-384:
 *     kernel = fftshift(irfft(exp(log_mag))) * window                    (filters.py:294-306)
 *     frames of N+hop-1 samples of the zero-padded excitation every hop  (filters.py:357-364)
 *     one cross-correlation per frame with that frame's kernel (grouped F.conv1d, filters.py:371-383)
 *
 *   N = 2*(n_mag-1) taps, P = (N-1)/2,  nfr = min((T + 2P - (N+hop-1))/hop + 1, F) frames,
 *   y[b, f*hop+n] = sum_{k<N} xp[b, f*hop+n+k] * kernel[b,f,k],  xp = ex zero-padded by P both sides,
 *   output length nfr*hop (golf_ltv_fir_frames_length; -1 if T is shorter than one frame span).
 *
 * The inverse real FFT of a zero-phase spectrum is a cosine transform: a dense (B*F, n_mag) x (n_mag, n_mag)
 * contraction, run on the matrix cores in exact fp32.  Its constant matrix (`basis`, both orientations) is built
 * once per n_mag by golf_zero_phase_fir_basis_f32 into a caller-owned buffer of golf_zero_phase_fir_basis_bytes().
 * Kernel rows live in a (B*F, row_stride) buffer, row_stride = golf_zero_phase_fir_row_stride(n_mag) >= N floats
 * (taps [N, row_stride) are written as zeros); golf_ltv_fir_frames_* accept any per-frame kernels in that layout.
 * ------------------------------------------------------------------------------------------- */
int golf_zero_phase_fir_row_stride(int n_mag);
size_t golf_zero_phase_fir_basis_bytes(int n_mag);
int golf_zero_phase_fir_basis_f32(int n_mag, void* basis, size_t basis_bytes, void* stream);

/* log_mag (G, n_mag), window (N) -> kern (G, row_stride);  G = B*F */
int golf_zero_phase_fir_kernels_f32(const float* log_mag, const float* window, const void* basis, float* kern,
                                    int G, int n_mag, void* stream);
/* g_kern (G, row_stride) -> g_log_mag (G, n_mag): the adjoint of window * fftshift * irfft, times exp(log_mag) */
int golf_zero_phase_fir_kernels_bwd_f32(const float* g_kern, const float* log_mag, const float* window,
                                        const void* basis, float* g_log_mag, int G, int n_mag, void* stream);

int golf_ltv_fir_frames_length(int T, int F, int N, int hop);
/* ex (B, >=T) stride ex_stride; kern (B*F, kern_row_stride), zero in [N, ceil4(N)); y (B, nfr*hop) stride y_stride.
 * frame0 (normally 0): output frame f is filtered with kernel row f + frame0, nfr = min(frames in T, F - frame0) —
 * what the sample-wise variant (LTVZeroPhaseFIRFilterPrecise, models/filters.py:286-337: kernels interpolated between
 * frames f and f+1) needs for its second term. */
int golf_ltv_fir_frames_fwd_f32(const float* ex, int64_t ex_stride, const float* kern, int kern_row_stride, float* y,
                                int64_t y_stride, int B, int T, int F, int N, int hop, int frame0, void* stream);
/* gy (B, nfr*hop).  g_ex (B,T) and g_kern (B*F, kern_row_stride; rows of unused frames and the padding taps [N, kern_row_stride) zeroed) are fully
 * overwritten; either may be NULL to skip it.  Requires hop % 4 == 0. */
int golf_ltv_fir_frames_bwd_f32(const float* gy, int64_t gy_stride, const float* ex, int64_t ex_stride,
                                const float* kern, int kern_row_stride, float* g_ex, int64_t g_ex_stride,
                                float* g_kern, int B, int T, int F, int N, int hop, int frame0, void* stream);

/* ---------------------------------------------------------------------------------------------
 * f-2: LTI FIR shared by the whole batch — the room filter after the end filter.
 * Replaces LTIAcousticFilter.forward, models/filters.py:426-449 (pad + F.conv1d with one learnable kernel):
 *     y[b,t] = sum_{n<ntaps} taps[n] * ex[b, t - lead + n],   ex = 0 outside [0,T)
 *   (the reference's y = ex + conv(pad(ex[:, :-1], (K,0)), kernel) is taps = [kernel, 1], lead = K).
 *   ntaps must be a multiple of 4 (pad with zero taps), 0 <= lead < ntaps.
 *   The adjoint w.r.t. ex is the same call with the taps reversed and lead' = ntaps-1-lead; ABI 5: ntaps < 0 applies the
 *   |ntaps| taps in reverse order, so the adjoint needs no flipped copy of them.
 * golf_lti_fir_taps_grad_f32: g_taps[n] = sum_{b,t} gy[b,t] * ex[b, t - lead + n]  (deterministic two-stage sum).
 * ------------------------------------------------------------------------------------------- */
int golf_lti_fir_f32(const float* ex, int64_t ex_stride, const float* taps, int ntaps, int lead, float* y,
                     int64_t y_stride, int B, int T, void* stream);
size_t golf_lti_fir_taps_grad_workspace_bytes(int B, int T, int ntaps);
int golf_lti_fir_taps_grad_f32(const float* gy, int64_t gy_stride, const float* ex, int64_t ex_stride, float* g_taps,
                               int ntaps, int lead, int B, int T, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a-12: filtered-noise-band generator.  Replaces NoiseBand.forward, models/noise.py:114-124:
 *     out[b,t] = sum_k noise_bands[k][(t + offsets[b,k]) mod Lb] * up(exp(log_gain))[b,t,k]
 *   noise_bands (K, Lb) one loopable period per band (Lb a power of two; built at init by the host module as
 *   models/noise.py:141-213 does), offsets (B,K) int32 in [0,Lb) (the random start per utterance and band),
 *   log_gain (B,F,K) at hop `hop`, out (B,T), T <= (F-1)*hop+1.  The (B,K,T) gather and the (B,T,K) upsampled gains of
 *   the reference are never formed.  Backward w.r.t. log_gain (g_log_gain (B,F,K) fully overwritten).
 * ------------------------------------------------------------------------------------------- */
size_t golf_noise_band_workspace_bytes(int B, int F, int K);
int golf_noise_band_fwd_f32(const float* noise_bands, int Lb, const int* offsets, const float* log_gain, int F, int hop,
                            float* out, int64_t out_stride, int B, int T, int K, void* stream);
int golf_noise_band_bwd_f32(const float* g_out, int64_t g_out_stride, const float* noise_bands, int Lb,
                            const int* offsets, const float* log_gain, int F, int hop, float* g_log_gain, int B, int T,
                            int K, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a-11: harmonic oscillator bank — the source of the DDSP / NHV / WORLD / MLSA / SawSing / PULF baselines.
 * Replaces HarmonicOscillator.forward, models/synth.py:403-446, and what AdditiveSynthesizer (:449-468),
 * SawToothOscillator (:486-504) and AdditivePulseTrain (:526-547) feed it:
 *     p = up(phase)  [cycles/sample, linear upsampling from phase_hop];   Phi = inclusive cumsum of p
 *     out[t] = sum_{h=1..H} [h*p(t) < 0.5] * amp(t,h) * sin(2 pi h Phi(t))
 *     amp(t,h) = up(A)[t,h] * up(tscale)[t] * hscale[h];  A (B,Fa,H) at amp_hop, tscale (B,Fs) at ts_hop, hscale (H):
 *     each may be NULL (= 1).  Tout = min of the upsampled lengths ((n-1)*hop+1) of phase and the given factors.
 *   The (B,T,H) tensors of the reference are never formed; the phase is exact (64-bit fixed point, as in the
 *   wavetable oscillator), sin(h theta) by a rotation recurrence re-anchored from the exact phase every 32 harmonics.
 * Backward w.r.t. A (g_amp (B,Fa,H) fully overwritten); golf_harmonic_osc_dphase_f32 serves the gradient w.r.t. the phase.
 *   Optional phase terms (ABI 3; models/synth.py:434-440): harmonic h runs at h * (Phi + up(phase_offset)) + initial_phase[b,h];
 *   phase_offset (B,Fo) cycles at hop po_hop (linear upsampling, must cover Tout samples), initial_phase (B,H) cycles;
 *   either may be NULL.  d out / d phase_offset(t) is what golf_harmonic_osc_dphase_f32 returns.
 * ------------------------------------------------------------------------------------------- */
/* Fa = amplitude frames (0 if amp is NULL); sized for the forward and the backward */
size_t golf_harmonic_osc_workspace_bytes(int B, int Tp, int phase_hop, int Fa, int H);
int golf_harmonic_osc_fwd_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                              const float* amp, int Fa, int amp_hop, const float* tscale, int Fs, int ts_hop,
                              const float* hscale, int H, float* out, int64_t out_stride, int B, int Tout,
                              void* ws, size_t ws_bytes, void* stream,
                              const float* phase_offset, int Fo, int po_hop, const float* initial_phase);
/* d out / d Phi(t) (Phi = the running phase in cycles): 2 pi sum_h [h p < 0.5] amp(t,h) h cos(2 pi h Phi(t)); the
 * gradient w.r.t. the phase input is the transposed upsampling of the reverse cumulative sum of g_out * this (host). */
int golf_harmonic_osc_dphase_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                              const float* amp, int Fa, int amp_hop, const float* tscale, int Fs, int ts_hop,
                              const float* hscale, int H, float* out, int64_t out_stride, int B, int Tout,
                              void* ws, size_t ws_bytes, void* stream,
                              const float* phase_offset, int Fo, int po_hop, const float* initial_phase);
int golf_harmonic_osc_bwd_amp_f32(const float* g_out, int64_t g_out_stride, const float* phase, int64_t phase_stride,
                                  int Tp, int phase_hop, int Fa, int amp_hop, const float* tscale, int Fs, int ts_hop,
                                  const float* hscale, int H, float* g_amp, int B, int Tout,
                                  void* ws, size_t ws_bytes, void* stream,
                                  const float* phase_offset, int Fo, int po_hop, const float* initial_phase);

/* ---------------------------------------------------------------------------------------------
 * (e) multi-GPU: push-based exchange of the synthesised audio (north_star: "shards independent utterances across the 8
 * MI355X ... all-gathering only the synthesized audio").  The reference has no counterpart (it is single-device:
 * autoencode.py / test_rtf.py); what these replace is the torch.distributed.all_gather_into_tensor that
 * golf_amd.dist.gather_audio issues.  xGMI is point to point, so instead of a ring collective ONE kernel stores a
 * step's (rows, T) block into every peer's receive buffer (7 links at once, source read once), then publishes a
 * sequence number with system scope; the consumer waits for it in-stream.
 *   golf_peer_alloc     fine-grained device memory, zero-filled (receive buffers and flags: pollable while kernels run)
 *   golf_peer_export    -> GOLF_PEER_HANDLE_BYTES opaque bytes (hipIpcMemHandle_t) to send to the other processes
 *   golf_peer_open/close   map / unmap another process's buffer
 *   golf_peer_store_f32    src (rows, T) -> dst[i] (rows, T) for i < n_dst <= GOLF_MAX_PEERS; dst = HOST array of device
 *                          pointers (already offset to this rank's slot in each peer's buffer)
 *   golf_peer_signal_u32   *flags[i] = seq for i < n (release, system scope; ordered after this stream's stores)
 *   golf_peer_wait_u32     blocks the STREAM until flags[i*stride] >= seq for all i < n (signed distance: sequence numbers
 *                          may wrap), or sets *status (device int) to 1 + i after timeout_us -- never hangs the GPU
 * Used by golf_amd.dist.PeerStoreGather (bench.py --gather-mode peer-store).  Exercised with 2 processes on one GPU;
 * not yet measured on a multi-GPU node -- RCCL's all-gather remains the default exchange. */
#define GOLF_MAX_PEERS 16
#define GOLF_PEER_HANDLE_BYTES 64
int golf_peer_alloc(size_t bytes, void** ptr);
int golf_peer_free(void* ptr);
int golf_peer_export(void* ptr, void* handle64);
int golf_peer_open(const void* handle64, void** ptr);
int golf_peer_close(void* ptr);
int golf_peer_store_f32(const float* src, int64_t src_stride, int rows, int T, void* const* dst, int64_t dst_stride,
                        int n_dst, void* stream);
int golf_peer_signal_u32(void* const* flags, int n, uint32_t seq, void* stream);
int golf_peer_wait_u32(const uint32_t* flags, int n, int stride, uint32_t seq, int64_t timeout_us, int* status,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GOLF_AMD_H */
