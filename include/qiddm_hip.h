/*
 * qiddm_hip.h -- C ABI of the MI355X (gfx950) statevector engine behind the
 * QIDDM quantum layers.
 *
 * The reference (aaai2026/QIDDM) has no FFI for this path: its boundary is the
 * Python call `self.qnode(inputs[, weights])` on a PennyLane QNode
 * (reference nn/qdense.py:26-38,58 ; :237-247,279 ; :406-420,465 ; :1586-1600,1633 ;
 * nn/qconv.py:39-47) whose arithmetic lives in the third-party simulators
 * PennyLane 0.29.0 `default.qubit.torch` / PennyLane-Lightning 0.30.0
 * `lightning.qubit` (reference requirements.txt:44-45).  This header is the
 * drop-in for that simulator: every entry point below is what a QNode
 * execution (forward, parameter-shift sweep, adjoint backward) binds to.  The
 * Python binding (ctypes) is qiddm_amd/_capi.py; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t
 *     passed as void* (NULL = the null stream).
 *   - every pointer except `circ` is a DEVICE pointer owned by the caller, who
 *     keeps it alive until the stream has been synchronised.  No allocation,
 *     no synchronisation and no host<->device copies happen inside any call,
 *     so every call is hipGraph-capturable, re-entrant and thread safe.
 *   - return value: QIDDM_OK (0) or a negative qiddm_status; a human readable
 *     reason is available from qiddm_last_error() (thread local).
 *   - wire w is bit (n-1-w) of the amplitude index (wire 0 = most significant),
 *     the order qml.probs(wires=range(n)) reports.
 *
 * Circuit family (covers every `_circuit` of reference nn/qdense.py and the
 * exported QConv2d of nn/qconv.py):
 *
 *   for round in 0..n_rounds-1:                 chained QNode calls, nn/qdense.py:464-465, 1631-1635
 *     state <- AmplitudeEmbedding(x + enc_offset, pad_with, normalize)   (QIDDM_ENC_AMPLITUDE)
 *              or |0...0>
 *     for block in 0..n_blocks-1:                data re-uploading, nn/qdense.py:424-428
 *       QIDDM_ENC_RZ : RZ(enc_scale * x_j) on every wire j
 *       QIDDM_ENC_RY : RY(enc_scale * x_j) on every wire j, block 0 only (qml.AngleEmbedding)
 *       QIDDM_ENC_RY_BLOCKS : the same RY layer in front of every block
 *       StronglyEntanglingLayers(angles[round][block] : (sel_layers, n, 3), imprimitive)
 *     out <- probs (B, 2^n)  |  <Z_i> (B, n)
 *     x   <- out[:, 0:n]     (input of the next round)
 */
#ifndef QIDDM_HIP_H
#define QIDDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QIDDM_ABI_VERSION 1
#define QIDDM_MAX_QUBITS_FUSED 10 /* one wavefront owns the whole slab in registers */
#define QIDDM_MAX_QUBITS 16       /* 11..16: one workgroup per sample, tiled passes over a workspace slab */

typedef enum qiddm_status {
  QIDDM_OK = 0,
  QIDDM_ERR_INVALID = -1,     /* bad argument (null pointer, size, enum)            */
  QIDDM_ERR_UNSUPPORTED = -2, /* valid request this build cannot run (e.g. n > max)  */
  QIDDM_ERR_LAUNCH = -3       /* HIP reported a launch failure                       */
} qiddm_status;

typedef enum qiddm_encoding {
  QIDDM_ENC_NONE = 0,
  QIDDM_ENC_AMPLITUDE = 1, /* qml.AmplitudeEmbedding  (nn/qdense.py:41-43, nn/qconv.py:52-54) */
  QIDDM_ENC_RZ = 2,        /* qml.RZ(inputs[:, j], wires=j) (nn/qdense.py:253, 427, 1408)     */
  QIDDM_ENC_RY = 3,        /* qml.AngleEmbedding(rotation="Y"): once, before block 0 (nn/qdense.py:166-168) */
  QIDDM_ENC_RY_BLOCKS = 4  /* qml.RY(inputs[j], wires=j) re-uploaded in every block (nn/qdense.py:600-602) */
} qiddm_encoding;

typedef enum qiddm_imprimitive {
  QIDDM_IMP_CNOT = 0, /* StronglyEntanglingLayers default (nn/qdense.py:44-46)              */
  QIDDM_IMP_CZ = 1    /* imprimitive=qml.ops.CZ (nn/qdense.py:263, 428)                      */
} qiddm_imprimitive;

typedef enum qiddm_measure {
  QIDDM_MEAS_PROBS = 0, /* qml.probs(wires=range(n))                 (nn/qdense.py:47)       */
  QIDDM_MEAS_EXPZ = 1   /* [qml.expval(qml.PauliZ(i)) for i in ...]  (nn/qdense.py:264)      */
} qiddm_measure;

typedef enum qiddm_dtype {
  QIDDM_F32 = 0, /* complex64 state, float I/O  (the production path)                        */
  QIDDM_F64 = 1  /* complex128 state, double I/O (the reference's own precision, F5)         */
} qiddm_dtype;

typedef struct qiddm_circuit {
  int32_t n_qubits;    /* 1 .. qiddm_max_qubits() (= 16)                                     */
  int32_t encoding;    /* qiddm_encoding                                                     */
  int32_t imprimitive; /* qiddm_imprimitive                                                  */
  int32_t measure;     /* qiddm_measure                                                      */
  int32_t n_rounds;    /* N >= 1                                                             */
  int32_t n_blocks;    /* L >= 1                                                             */
  int32_t sel_layers;  /* S >= 1  (ranges r_s = (s mod (n-1)) + 1 restart in every block)    */
  int32_t n_features;  /* columns of `inputs` that are read: <= 2^n (amplitude), >= n (RZ/RY) */
  int32_t dtype;       /* qiddm_dtype of inputs / outputs / gate table                       */
  int32_t reserved;    /* must be 0                                                          */
  double enc_scale;    /* multiplies the input angles (1.0; pi/2 in nn/qdense.py:2215)       */
  double enc_offset;   /* added to the features before amplitude embedding (nn/qconv.py:78)  */
  double pad_with;     /* amplitude-embedding pad constant (0.1 qdense, 0.5 qconv)           */
} qiddm_circuit_t;

/* ---- introspection ------------------------------------------------------- */
int qiddm_abi_version(void);
int qiddm_max_qubits(void);
const char *qiddm_last_error(void);

/* Diagnostics only (tools/stamp_*.py): a device buffer of n_words 64-bit words into which workgroup 0 of the
 * dense / quantum-convolution kernels writes s_memtime phase stamps (they serialise the kernel: never set it in a
 * timed run).  The pointer is checked (device memory, the words inside one allocation); NULL / 0 clears it.  There
 * is no environment-variable form: a stale value inherited from a shell would be a wild GPU write.          */
int qiddm_set_stamp_buffer(void *device_ptr, int64_t n_words);

/* number of Rot gates = n_rounds*n_blocks*sel_layers*n_qubits (= angles / 3)            */
int64_t qiddm_num_rot_gates(const qiddm_circuit_t *circ);
/* gate applications per sample per forward, counted as SURVEY.md section 8a             */
int64_t qiddm_gate_count(const qiddm_circuit_t *circ);
/* elements (of circ->dtype) in the gate table qiddm_prepare_gates writes:
 * num_rot_gates * 7 variants * 8 reals, plus (n <= 10) the folded per-layer tables the forward of a
 * CZ circuit runs on: per layer n (cos, sin)(theta/2) pairs and 2^min(n,6) + 2^max(n-6,0) phases
 * exp(i sum +-(phi^l + omega^(l-1))/2) -- RZ(phi) RZ(omega) and the CZ ring between two RY layers
 * are one diagonal; for n = 11..16 CZ circuits with no / RZ encoding, per layer and wire the pairs
 * (cos, sin)(theta/2) and (cos, sin)((phi^l + omega^(l-1))/2) the pass-structured kernels run on      */
int64_t qiddm_gate_table_elems(const qiddm_circuit_t *circ);
/* replicas of a full parameter-shift sweep: 6*num_rot_gates (+ 2*n_blocks*n for the
 * input angles of RZ/RY encodings)                                                       */
int64_t qiddm_num_shift_replicas(const qiddm_circuit_t *circ, int with_inputs);
/* bytes of device workspace qiddm_forward (n_replicas = 0) / qiddm_forward_shifted need for this
 * circuit and batch: 0 for n <= 10 (register-resident slabs); for n = 11..16 2^n-amplitude slabs for
 * the concurrently resident workgroups (up to 1024: the sweeps of a 16-qubit shard are HBM traffic).  */
int64_t qiddm_workspace_bytes(const qiddm_circuit_t *circ, int64_t batch, int64_t n_replicas);

/* ---- gate table ----------------------------------------------------------
 * angles: (n_rounds, n_blocks, sel_layers, n, 3) float64, the weights tensor of
 * StronglyEntanglingLayers AFTER any host-side map (qw_map.tanh, torch.tanh).
 * Writes for every Rot gate the 2x2 matrix RZ(omega) RY(theta) RZ(phi) (variant 0)
 * and its six +-pi/2 parameter shifts (variants 1..6 = phi+,phi-,theta+,theta-,
 * omega+,omega-), evaluated in float64 and rounded once to circ->dtype.
 * Replaces qml.StronglyEntanglingLayers' decomposition into qml.Rot
 * [PennyLane 0.29.0 templates/layers/strongly_entangling.py].                            */
int qiddm_prepare_gates(const qiddm_circuit_t *circ, const double *angles, void *gate_table,
                        void *stream);

/* ---- forward ----------------------------------------------------------------
 * Replaces one (n_rounds == 1) or N chained QNode executions
 * `self.qnode(inputs, weights)` -> default.qubit.torch / lightning.qubit
 * (reference nn/qdense.py:58, 279, 465, 1439, 1633).
 * inputs: (batch, in_ld) row-major, first n_features columns read (may be NULL
 *         for QIDDM_ENC_NONE).  out: (batch, out_ld) rows of 2^n probabilities or
 *         n expectation values.  n <= 10: one wavefront owns one sample's slab in
 *         registers for the whole circuit.  n = 11..16: one workgroup per sample, slab in
 *         `workspace` (>= qiddm_workspace_bytes(circ, batch, 0) bytes; may be NULL for
 *         n <= 10), tiled passes (qiddm_amd/csrc/qsim_tiled.h).                            */
int qiddm_forward(const qiddm_circuit_t *circ, const void *inputs, int64_t batch, int64_t in_ld,
                  const void *gate_table, void *out, int64_t out_ld, void *workspace,
                  int64_t workspace_bytes, void *stream);

/* ---- parameter-shift sweep ---------------------------------------------------
 * Replaces PennyLane's diff_method="parameter-shift" executions configured at
 * reference nn/qdense.py:246, 1400, 1596: re-invokes the forward kernel for
 * replicas [first_replica, first_replica + n_replicas) of the shift schedule and
 * contracts each replica's output with the upstream gradient on the fly:
 *     dots[r - first_replica][b] = sum_k grad_out[b][k] * out_r[b][k]
 * Replica ids: 6*g + v (v = 0..5 -> phi+,phi-,theta+,theta-,omega+,omega-) for Rot
 * gate g, then 6*G_rot + 2*(block*n + wire) + (0:+, 1:-) for the input angle of
 * `wire` re-uploaded in `block`.  d/dtheta = (dots[+] - dots[-]) / 2 summed over
 * the batch (weights) or per sample (inputs; times enc_scale).  n_rounds must be 1.      */
/* The same forward for the PROBABILITY nets with their `_post_process` fused into the store (reference
 * nn/qdense.py:49-54, 443-448: `clamp(probs[:, :pixels] * pixels, 0, 1)` on the float64 the QNode returns):
 * out (batch, post_cols) float64 = clamp((double)p[:, :post_cols] * post_scale, 0, 1) -- bit-identical to converting the
 * probabilities to float64 first; the (batch, 2^n) matrix is never written.  measure = probs, n <= 10.           */
int qiddm_forward_post(const qiddm_circuit_t *circ, const void *inputs, int64_t batch, int64_t in_ld,
                       const void *gate_table, double *out, int64_t out_ld, int32_t post_cols, double post_scale,
                       void *stream);

int qiddm_forward_shifted(const qiddm_circuit_t *circ, const void *inputs, int64_t batch,
                          int64_t in_ld, const void *gate_table, const void *grad_out,
                          int64_t g_ld, int64_t first_replica, int64_t n_replicas, void *dots,
                          void *workspace, int64_t workspace_bytes, void *stream);

/* ---- adjoint (reverse-mode) backward ----------------------------------------------------
 * Replaces what diff_method="backprop" computes in the reference (torch autograd through
 * default.qubit.torch, nn/qdense.py:37, 419; nn/qconv.py:46) for ONE QNode round (n_rounds == 1,
 * n_qubits <= 10):  given grad_out = dL/d(out) of qiddm_forward it returns
 *   k_partials : (n_partials, n_rot, 8) of circ->dtype -- per-workgroup partial sums over the batch of
 *                K_{ab} = sum conj(lambda_a) psi_b for every Rot gate (a,b in {0,1}, as re/im pairs in the
 *                order K00 K01 K10 K11); the caller sums them over dim 0 and contracts with the analytic
 *                dRot/d(phi,theta,omega):  dL/dangle = 2 Re sum_ab (dU/dangle)_ab K_ab.
 *                n_partials = qiddm_adjoint_partials(circ, batch).  For CZ circuits without RY data encoding and
 *                n <= 9 the slabs (same size) hold per-layer sums of d/dtheta and d/d(phi^l + omega^(l-1)) instead
 *                (folded reverse sweep): treat them as opaque and hand them to qiddm_adjoint_finalize.
 *   grad_inputs: (batch, gin_ld) dL/d(inputs): n columns for RZ / RY encodings, n_features columns for
 *                the amplitude embedding (gradient through pad + normalise); may be NULL.
 * Cost: about five forward passes, independent of the number of parameters.                     */
int64_t qiddm_adjoint_partials(const qiddm_circuit_t *circ, int64_t batch);
int qiddm_backward_adjoint(const qiddm_circuit_t *circ, const void *inputs, int64_t batch,
                           int64_t in_ld, const void *gate_table, const void *grad_out, int64_t g_ld,
                           void *k_partials, void *grad_inputs, int64_t gin_ld, void *stream);
/* n = 11..16: the same gradient with psi and lambda in a per-workgroup slab pair of `workspace`
 * (qiddm_adjoint_workspace_bytes; 0 for n <= 10).  CZ circuits with no / RZ encoding and >= 2 layers: the
 * pass-structured reverse sweep (one sweep of both slabs per LAYER; the slabs of k_partials then hold per-layer
 * angle-gradient sums: opaque, hand them to qiddm_adjoint_finalize); everything else: one sweep per gate, K slabs. */
int64_t qiddm_adjoint_workspace_bytes(const qiddm_circuit_t *circ, int64_t batch);
int qiddm_backward_adjoint_wide(const qiddm_circuit_t *circ, const void *inputs, int64_t batch,
                                int64_t in_ld, const void *gate_table, const void *grad_out, int64_t g_ld,
                                void *k_partials, void *grad_inputs, int64_t gin_ld, void *workspace,
                                int64_t workspace_bytes, void *stream);
/* Sums the n_partials K slabs (fixed order: deterministic) and contracts them with the analytic
 * dRot/d(phi, theta, omega) in float64: grad_angles (n_rot, 3) float64 = dL/d(angles).              */
int qiddm_adjoint_finalize(const qiddm_circuit_t *circ, const double *angles, const void *k_partials,
                           int64_t n_partials, double *grad_angles, void *stream);

/* ---- fused dense-net forward ------------------------------------------------------
 * Replaces the whole forward of the reference's linear_down -> quantum rounds -> linear_up
 * nets in one launch: QNN_noise.forward / QNN.forward (reference nn/qdense.py:267-289,
 * 346-368) and QIDDM_LL_noise.forward (:1620-1642), i.e.
 *     h   = x @ w_down^T + b_down                      (batch, n)        float64
 *     ev  = circuit(h)  (circ: QIDDM_ENC_RZ, QIDDM_MEAS_EXPZ, any n_rounds/n_blocks/sel_layers)
 *     net = ev @ w_up^T + b_up                         (batch, out_features)  float64
 * and, with post_mode = 1, the "noise"-goal update of the sampling loop on top of it
 * (reference src/models.py:130-134):  y = clamp(x - (net - 0.5) * 0.1 * noise_factor, 0, 1)
 * (post_mode = 0: y = net).  The classical glue runs in float64 as in the reference, the
 * statevector in circ->dtype.  `angles` is the raw (n_rounds, n_blocks, sel_layers, n, 3)
 * float64 weight tensor: the Rot matrices are built inside the launch (no gate table).
 * w_down: (n, in_features) row-major, w_up: (out_features, n) row-major (torch.nn.Linear
 * layout); b_down / b_up may be NULL.  y must not alias x.  post_mode = 1 needs
 * out_features == in_features.                                                           */
int qiddm_dense_forward(const qiddm_circuit_t *circ, const double *x, int64_t batch, int64_t x_ld,
                        int64_t in_features, const double *w_down, const double *b_down,
                        const double *angles, const double *w_up, const double *b_up,
                        int64_t out_features, int32_t post_mode, double noise_factor, double *y,
                        int64_t y_ld, void *stream);

/* ---- fused sampling loop -----------------------------------------------------------------
 * `n_steps` consecutive bodies of the reference's sampling loop (src/models.py:124-136)
 *     x <- net(x)                                        (post_mode 0, "data" goal)
 *     x <- clamp(x - (net(x) - 0.5) * 0.1 * noise_factor, 0, 1)   (post_mode 1, "noise" goal)
 * in ONE launch, net = linear_down -> circuit rounds -> linear_up as in qiddm_dense_forward.
 * y: (n_steps, batch, out_features) float64, y[s] = the image after step s+1 (row stride y_ld,
 * step stride y_step_stride).  n_steps > 1 needs out_features == in_features.  Four wavefronts own
 * one sample (qsim_quad.h): supported for 2 <= n_qubits <= 10 (below 8 every wavefront runs the circuit on its own copy of the state), QIDDM_IMP_CZ, QIDDM_ENC_RZ,
 * QIDDM_MEAS_EXPZ, features <= 2048; anything else returns QIDDM_ERR_UNSUPPORTED (loop over
 * qiddm_dense_forward instead).                                                               */
int qiddm_dense_sample(const qiddm_circuit_t *circ, const double *x, int64_t batch, int64_t x_ld,
                       int64_t in_features, const double *w_down, const double *b_down,
                       const double *angles, const double *w_up, const double *b_up,
                       int64_t out_features, int32_t post_mode, double noise_factor, int32_t n_steps,
                       double *y, int64_t y_ld, int64_t y_step_stride, const void *tables, void *stream);
/* The sampler runs on per-layer tables derived from `angles` (RY coefficients, folded RZ/CZ phases).  With
 * tables == NULL every launch rebuilds them in LDS; a caller that samples many times with the same weights
 * builds them once: qiddm_dense_sample_prepare writes qiddm_dense_sample_tables_bytes(circ) bytes (circ->dtype)
 * that stay valid until the angles change.                                                              */
int64_t qiddm_dense_sample_tables_bytes(const qiddm_circuit_t *circ);
int qiddm_dense_sample_prepare(const qiddm_circuit_t *circ, const double *angles, void *tables, void *stream);

/* ---- lean sampling loop of the 8- and 6-qubit dense nets (qsim_lean.h) --------------------------------------------
 * The same n_steps bodies of Diffusion.sample (post_mode / noise_factor as for qiddm_dense_sample, reference
 * src/models.py:127-134) on 8 or 6 wires / CZ rings / RZ encoding / <Z>, with every RY in tangent form (one fused
 * cross-lane multiply-add per gate).  post_mode 0 ("data" goal, x <- net(x)): linear_down is composed with linear_up
 * into an n x n map of the previous step's <Z> (W_down is only read for the first step, and not at all when the circuit
 * has one block per round: the data angles are then a global phase).  post_mode 1 ("noise" goal): the clamp breaks that
 * composition; the image and both linears' weights stay in registers across the steps.  Tables: qiddm_dense_sample_lean_tables_bytes(circ) bytes, written once per
 * weights by _prepare (angles AND the two linears); _check synchronises the stream and returns 1 when the tables are
 * usable -- max |tan(theta/2)| <= 16 over the simulated layers -- 0 when the caller must use qiddm_dense_sample, < 0
 * on error.  Results agree with qiddm_dense_sample to rounding (not bit for bit).                                  */
int64_t qiddm_dense_sample_lean_tables_bytes(const qiddm_circuit_t *circ);
int qiddm_dense_sample_lean_prepare(const qiddm_circuit_t *circ, const double *angles, const double *w_down,
                                    const double *b_down, const double *w_up, const double *b_up, int64_t features,
                                    void *tables, void *stream);
int qiddm_dense_sample_lean_check(const qiddm_circuit_t *circ, const void *tables, void *stream);
int qiddm_dense_sample_lean(const qiddm_circuit_t *circ, const double *x, int64_t batch, int64_t x_ld,
                            int64_t features, const double *w_down, const double *b_down, const double *w_up,
                            const double *b_up, int32_t post_mode, double noise_factor, int32_t n_steps, double *y,
                            int64_t y_ld, int64_t y_step_stride, const void *tables, void *stream);

/* ---- fused quantum convolution -------------------------------------------------------
 * Replaces the (intended, SURVEY finding F3) forward of the reference's exported QConv2d
 * (`_QConv2d_FAST`, nn/qconv.py:51-87) in one launch:
 *     torch.nn.Unfold(kernel, padding) -> + enc_offset (0.1) -> AmplitudeEmbedding(pad_with 0.5,
 *     normalize) -> StronglyEntanglingLayers(angles, CNOT) -> probs -> * 2^n / 2 -> clamp [0,1] ->
 *     [:, ::2][:, :out_channels] -> (batch, out_channels, H_out, W_out)
 * x: (batch, in_channels, H, W) float64 contiguous; y: (batch, out_channels, H_out, W_out) float64
 * with H_out = H + 2*pad_h - kh + 1.  circ: QIDDM_ENC_AMPLITUDE, QIDDM_MEAS_PROBS, n_rounds =
 * n_blocks = 1, n_features = in_channels*kh*kw, n_qubits <= 10.  `angles` is the (1,1,S,n,3) float64
 * tensor AFTER the qw_map.tanh map.  out_channels beyond 2^n / 2 do not exist (the reference's slice
 * silently returns fewer channels): they are rejected here.                                  */
int qiddm_qconv_forward(const qiddm_circuit_t *circ, const double *x, int64_t batch,
                        int64_t in_channels, int64_t height, int64_t width, int64_t kh, int64_t kw,
                        int64_t pad_h, int64_t pad_w, const double *angles, int64_t out_channels,
                        double *y, void *stream);

/* Backward of qiddm_qconv_forward in two launches (replaces torch autograd through Unfold, default.qubit.torch
 * and the post-processing slices, reference nn/qconv.py:46, 58-87 with diff_method="backprop"): the adjoint
 * sweep runs one circuit per output pixel with the patch read from x and dL/dp read from grad_y
 * (B, out_channels, H_out, W_out) through the clamp / [::2] / [:out_channels] chain, never materialising the
 * (B H_out W_out, 2^n) probability gradient; then the per-pixel feature gradients are folded back onto the image.
 *   gate_table : qiddm_prepare_gates of the (1,1,S,n,3) angles (after the tanh map), circ->dtype
 *   k_partials : (qiddm_adjoint_partials(circ, batch*H_out*W_out), n_rot, 8) circ->dtype -> qiddm_adjoint_finalize
 *   grad_features : scratch (batch*H_out*W_out, in_channels*kh*kw) circ->dtype; grad_x: (batch, C, H, W) float64.
 *                   Both NULL when the input needs no gradient.
 * n_qubits <= 10.                                                                                          */
int qiddm_qconv_backward(const qiddm_circuit_t *circ, const double *x, int64_t batch, int64_t in_channels,
                         int64_t height, int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                         const void *gate_table, const double *grad_y, int64_t out_channels, void *k_partials,
                         void *grad_features, double *grad_x, void *stream);

/* ---- fused training step (device-resident Diffusion step) ---------------------------------
 * Replaces, for nets of the linear_down -> circuit -> linear_up family (QNN_noise nn/qdense.py:267-289,
 * QIDDM_LL_noise :1620-1642), one call of Diffusion.run_training_step_data / _noise
 * (src/models.py:44-72 / :74-104) including its internal .backward(), with
 * add_normal_noise_multiple (src/noise.py:105-126) as the noising schedule:
 *     whole = x*(1-w_t) + noise*w_t, clamp      t = 0..tau        (never materialised)
 *     noisy = whole[:, 1:], clean = whole[:, :-1]                 rows i = b*tau + (t-1)
 *     out   = W_up <Z>(circuit(W_down noisy + b_down)) + b_up
 *     goal 0 ("data"):  loss = mean((out - clean)^2)
 *     goal 1 ("noise"): loss = mean(((out - 0.5)*0.1 - (noisy - clean))^2)
 * Outputs: the scalar loss and d loss / d parameter for every parameter (overwritten, not accumulated).
 * train_quantum = 0 reproduces the reference as written (SURVEY finding F1: the circuit output is
 * detached, only linear_up receives a gradient; g_w_down/g_b_down/g_angles are not touched);
 * train_quantum = 1 differentiates the circuit by the adjoint method (what diff_method="backprop"
 * yields in the reference).  recon / elem_loss (optional, (batch*tau, pixels)) are what the verbose call
 * returns: the reconstruction (goal 1: clamp(noisy - predicted, 0, 1)) and the element-wise loss.
 * circ: QIDDM_MEAS_EXPZ, any angle encoding or none, n_qubits <= 10, any n_rounds.
 * schedule: (tau+1) float32 weights with schedule[0] = 0; noise: float32 as the reference draws it.
 * All sums have a fixed order: bit-reproducible.  Three launches on `stream`, no host synchronisation. */
typedef struct qiddm_train_args {
  const double *x;        /* (batch, pixels) float64, row stride x_ld                      */
  float *noise;           /* (batch, pixels) float32, row stride noise_ld: the N(0.5, 0.2) field -- read,
                           * or, when rng_state != NULL, generated in the launch and written here        */
  const float *schedule;  /* (tau + 1)                                                     */
  int64_t x_ld, noise_ld, batch;
  int32_t pixels, tau, goal, train_quantum;
  const double *w_down, *b_down;   /* (n, pixels), (n) or NULL                              */
  const double *angles;            /* (n_rounds, n_blocks, sel_layers, n, 3)                */
  const double *w_up, *b_up;       /* (pixels, n), (pixels) or NULL                         */
  double *loss;                    /* (1)                                                   */
  double *g_w_down, *g_b_down, *g_angles, *g_w_up, *g_b_up;
  double *recon, *elem_loss;       /* optional                                              */
  uint64_t *rng_state;             /* optional: device {seed, offset}; Philox4x32-10 + Box-Muller per element,
                                    * offset advanced by one per call (graph-replayable)                 */
} qiddm_train_args_t;

/* bytes of scratch qiddm_train_step needs (negative: error code) */
int64_t qiddm_train_workspace_bytes(const qiddm_circuit_t *circ, int64_t batch, int32_t pixels, int32_t tau);
int qiddm_train_step(const qiddm_circuit_t *circ, const qiddm_train_args_t *args, void *workspace,
                     int64_t workspace_bytes, void *stream);

/* ---- Adam for every parameter tensor in one launch ------------------------------------------------
 * torch.optim.Adam as the reference harness constructs it (src/mnist_exm.py:170; betas, eps, weight_decay
 * passed explicitly; no amsgrad, no maximize):
 *     step += 1;  g += weight_decay * p;  m.lerp_(g, 1-beta1);  v = beta2 v + (1-beta2) g^2
 *     p -= lr / (1-beta1^step) * m / (sqrt(v) / sqrt(1-beta2^step) + eps)
 * step: per-tensor device int64 counter (torch keeps one per parameter; read, then advanced by one at the
 * end of the call); sync: device uint32 scratch, zero before the first call, left zero.  dtype per tensor QIDDM_F32/QIDDM_F64
 * (grad and moments have the parameter's dtype; the arithmetic is float64).  Safe to record into a HIP graph. */
typedef struct qiddm_adam_tensor {
  void *param;
  const void *grad;
  void *exp_avg, *exp_avg_sq;
  int64_t *step;
  int64_t numel;
  int32_t dtype, reserved;
} qiddm_adam_tensor_t;
int qiddm_adam_step(const qiddm_adam_tensor_t *tensors, int32_t n_tensors, double lr, double beta1,
                    double beta2, double eps, double weight_decay, uint32_t *sync, void *stream);

/* ---- eval-mode quantum convolution: circuit unitary + GEMM on the matrix cores ------------------------
 * Reference nn/qconv.py:92-126: `_QConv2d_FAST.train(False)` caches `sample_matrix = qml.matrix(SEL(pi*tanh(w)))`
 * and evaluates AmplitudeEmbedding -> QubitUnitary(sample_matrix) -> probs.
 *
 * qiddm_circuit_unitary: u (D, D) complex float64, row-major, interleaved (re, im): u[k][j] = <k|U|j> for the
 * weight-only circuit StronglyEntanglingLayers(angles (1,1,S,n,3), imprimitive) -- circ->encoding / measure
 * are ignored; n_rounds = n_blocks = 1, n_qubits <= 10.  Wire 0 is the most significant bit of k and j
 * (qml.matrix(..., wire_order=range(n))).
 *
 * qiddm_qconv_unitary_forward: the same function of (x, weights) as qiddm_qconv_forward, computed as the
 * implicit-im2col GEMM  (batch Ho Wo) x (C kh kw)  times  (C kh kw) x 2 C_out  on v_mfma_f32_32x32x2_f32
 * (float32 products and sums) with the normalise / |.|^2 / scale / clamp epilogue; x, y float64 as there.
 * workspace: qiddm_qconv_unitary_workspace_bytes (packed operand; rewritten by every call).
 * Two neighbours of the convolution in `unet_simple` (reference nn/unet_simple.py:9-18, 40-49) can ride along:
 *   upsample2x != 0: x is (batch, C, height, width) and the convolution reads its
 *       torch.nn.Upsample(scale_factor=2, mode="bilinear") (align_corners=False) -- the `up_conv` pair;
 *   bn != NULL: eval-mode BatchNorm2d on the output, (y - running_mean) / sqrt(running_var + eps) * weight + bias
 *       (weight / bias may be NULL) -- the [QConv2d, BatchNorm2d] pair.                                */
typedef struct qiddm_batchnorm {
  const double *weight, *bias, *running_mean, *running_var; /* (out_channels) each */
  double eps;
} qiddm_batchnorm_t;
int qiddm_circuit_unitary(const qiddm_circuit_t *circ, const double *angles, double *u, void *stream);
/* n <= 12 (C4's 12-qubit layers): writes the TRANSPOSE, ut[j][k] = <k|U|j> (each column of U contiguous: the
 * workgroup of column j evolves |j> in place in row j, no workspace).  Pass u_transposed = 1 below.          */
int qiddm_circuit_unitary_wide(const qiddm_circuit_t *circ, const double *angles, double *ut, void *stream);
/* The forward first packs the rows of u it needs (and the folded BatchNorm) into `workspace`.  u == NULL skips that launch:
 * `workspace` must then still hold the packing of an earlier call for the same unitary, batch norm and geometry -- an
 * eval-mode layer whose weights have not changed keeps its own workspace and packs once.                          */
int64_t qiddm_qconv_unitary_workspace_bytes(int32_t n_qubits, int64_t in_channels, int64_t kh, int64_t kw,
                                            int64_t out_channels);
int qiddm_qconv_unitary_forward(int32_t n_qubits, const double *u, const double *x, int64_t batch,
                                int64_t in_channels, int64_t height, int64_t width, int64_t kh, int64_t kw,
                                int64_t pad_h, int64_t pad_w, int64_t out_channels, int32_t upsample2x,
                                const qiddm_batchnorm_t *bn, int32_t u_transposed, double *y, void *workspace,
                                int64_t workspace_bytes, void *stream);

/* ---- training-mode BatchNorm2d, float64 NCHW ------------------------------------------------------------
 * The `torch.nn.BatchNorm2d` behind every quantum convolution of `unet_simple` (reference
 * nn/unet_simple.py:9-18, 30-39) in training mode: per-channel batch statistics (biased variance for the
 * transform, unbiased for the running estimate), running_mean / running_var moved by `momentum` in place
 * (either may be NULL), y = (x - mean) / sqrt(var + eps) * weight + bias (weight / bias may be NULL).
 * save_mean / save_invstd (channels) feed the backward: grad_x (may be NULL), grad_weight, grad_bias (may be
 * NULL).  x: (batch, channels, hw).  Two launches per direction, sums in a fixed order (deterministic).
 * workspace: qiddm_batchnorm_workspace_bytes, rewritten by every call.                                     */
int64_t qiddm_batchnorm_workspace_bytes(int64_t batch, int64_t channels, int64_t hw);
int qiddm_batchnorm_train_forward(const double *x, int64_t batch, int64_t channels, int64_t hw,
                                  const double *weight, const double *bias, double *running_mean,
                                  double *running_var, double momentum, double eps, double *y, double *save_mean,
                                  double *save_invstd, void *workspace, int64_t workspace_bytes, void *stream);
/* the statistics half of the backward only: grad_weight / grad_bias (each may be NULL) and coef (3, channels) with
 * dL/dx = coef[0][c] grad_y + coef[1][c] x + coef[2][c] -- for a producer that applies it itself
 * (qiddm_qconv_train_backward_bn).  Same workspace as qiddm_batchnorm_backward.                               */
int qiddm_batchnorm_backward_stats(const double *x, const double *grad_y, int64_t batch, int64_t channels, int64_t hw,
                                   const double *weight, const double *save_mean, const double *save_invstd,
                                   double *grad_weight, double *grad_bias, double *coef, void *workspace,
                                   int64_t workspace_bytes, void *stream);
int qiddm_batchnorm_backward(const double *x, const double *grad_y, int64_t batch, int64_t channels, int64_t hw,
                             const double *weight, const double *save_mean, const double *save_invstd,
                             double *grad_x, double *grad_weight, double *grad_bias, void *workspace,
                             int64_t workspace_bytes, void *stream);

/* ---- bilinear x2, float64 (planes, H, W) -> (planes, 2H, 2W) -------------------------------------------
 * torch.nn.Upsample(scale_factor=2, mode="bilinear") in front of every `up_conv` (reference
 * nn/unet_simple.py:40-49) and its backward, for training (the eval route reads the x2 inside the GEMM).
 * ah (2H, H), aw (2W, W): the dense 1-D interpolation matrices (the caller derives them from torch's operator
 * applied to the identity, so the weights are torch's); forward y = ah x aw^T, backward grad_x = ah^T grad_y aw,
 * both as gathers over the <= 3 / 4 non-zero entries per row / column.                                      */
int qiddm_upsample2x_forward(const double *x, int64_t planes, int64_t height, int64_t width, const double *ah,
                             const double *aw, double *y, void *stream);
int qiddm_upsample2x_backward(const double *grad_y, int64_t planes, int64_t height, int64_t width,
                              const double *ah, const double *aw, double *grad_x, void *stream);

/* ---- the dense unitary route (A4 nets at inference) ----------------------------------------------------------------------
 * AmplitudeEmbedding -> weight-only layers -> probs (reference nn/qdense.py:40-47) does not depend on the data beyond the
 * embedded vector: with U = qiddm_circuit_unitary(weights), a batch is ONE real product  [Re a | Im a] = v^ [Re U^T | Im U^T]
 * (a plain library GEMM on the caller's side) between two elementwise kernels:
 *   qiddm_amp_embed_rows:  v (batch, 2^n) float32 = (x + offset | pad_with) / norm       (normalize=True)
 *   qiddm_prob_post:       out (batch, cols) float64 = clamp((Re^2 + Im^2) * scale, 0, 1) from (batch, 2 cols) float32
 * -- the move the reference itself makes for its eval-mode QConv2d (nn/qconv.py:96-113).                               */
int qiddm_amp_embed_rows(const double *x, int64_t batch, int64_t x_ld, int64_t features, int32_t n_qubits, double pad_with,
                         double offset, float *v, void *stream);
int qiddm_prob_post(const float *amplitudes, int64_t batch, int64_t cols, double scale, double *out, void *stream);

/* MaxPool2d(kernel_size=2, stride=2) of the UNets' down blocks (reference nn/unet.py:93-95), float64 (planes, H, W) ->
 * (planes, H/2, W/2), floor mode.  The backward recomputes the winner of every window from x (first maximum in row-major
 * order, as torch picks it) instead of keeping an index tensor and writes every element of grad_x once.         */
int qiddm_maxpool2_forward(const double *x, int64_t planes, int64_t height, int64_t width, double *y, void *stream);
int qiddm_maxpool2_backward(const double *x, const double *grad_y, int64_t planes, int64_t height, int64_t width,
                            double *grad_x, void *stream);

/* ---- quantum convolution backward through the circuit unitary (training) ----------------------------------
 * The circuit of QConv2d does not depend on the data (reference nn/qconv.py:51-56), so with U = U(weights):
 * a_mc = sum_j U[2c,j] v^_mj, y_mc = clamp(|a_mc|^2 D/2) for every output pixel m -- and the backward needs no
 * per-pixel circuit sweep:  with t_mc = dL/dy_mc D/2 where the clamp passes,
 *     dL/dv^_mj = 2 Re sum_c t_mc conj(a_mc) U[2c,j]   -> through the normalisation and the fold -> grad_x
 *     dL/dangle = 2 Re sum_c <e_2c| dU/dangle |h_c>,   h_c[j] = sum_m t_mc conj(a_mc) v^_mj
 * qiddm_qconv_train_backward computes a, dL/dv (stored transposed, (C kh kw, M) float32, then folded into grad_x
 * unless NULL) and partial sums of h:  h_partials (qiddm_qconv_train_partials(..., C kh kw), 2 row_channels,
 * C kh kw + 1) float32 with  h_c[j] = sum_p hp[p][c][j] - i sum_p hp[p][row_channels + c][j];  column C kh kw is the
 * value every pad column j >= C kh kw shares.
 *   rows: (C kh kw + 1, 2 row_channels) float32 -- rows[j][c] = Re U[2c,j], rows[j][row_channels + c] = Im U[2c,j],
 *         last row 0.5 sum_{j >= C kh kw} U[2c,j]; zero for c >= out_channels.  row_channels in {8, 16, 32}
 *         (C kh kw <= 510): wider layers return QIDDM_ERR_UNSUPPORTED (use qiddm_qconv_backward or library GEMMs).
 *         From 32 patch features on the three products run on the f32 matrix cores (v_mfma_f32_16x16x4_f32 /
 *         32x32x2_f32); numel(x) and numel(grad_y) must then be below 2^32 (32-bit element offsets).
 * qiddm_matrix_adjoint: K slabs (-> qiddm_adjoint_finalize) of 2 Re <lambda_s| dU/dangle |psi0_s> summed over
 * `count` (psi0, lambda) pairs of complex128 vectors (count, 2^n, interleaved); circ->dtype QIDDM_F64, gate table
 * of that dtype, 2 <= n_qubits <= 16.                                                                        */
/* the two small steps either side: `rows` from the unitary (qiddm_circuit_unitary[_wide]; u_transposed as there),
 * and h_partials -> the (out_channels, 2^n) complex128 start vectors psi0 = h_c and lambda = e_2c of
 * qiddm_matrix_adjoint                                                                                        */
int qiddm_qconv_train_rows(int32_t n_qubits, const double *u, int32_t u_transposed, int64_t features,
                           int64_t out_channels, int32_t row_channels, float *rows, void *stream);
int qiddm_qconv_train_vectors(int32_t n_qubits, const float *h_partials, int64_t n_partials, int64_t features,
                              int64_t out_channels, int32_t row_channels, double *psi0, double *lambda, void *stream);
/* the fold on its own: grad_x (batch, C, H, W) float64 from transposed feature gradients (C kh kw, batch Ho Wo)
 * float32 -- the transpose of torch.nn.Unfold as a deterministic gather (layers too wide for the thin-product
 * kernel compute the feature gradients with library GEMMs, qiddm_amd/circuit.py)                              */
int qiddm_qconv_fold_features(const float *grad_features_t, int64_t batch, int64_t in_channels, int64_t height,
                              int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w, double *grad_x,
                              void *stream);
int64_t qiddm_qconv_train_partials(int64_t batch, int64_t height_out, int64_t width_out, int64_t features);
int qiddm_qconv_train_backward(int32_t n_qubits, const double *x, int64_t batch, int64_t in_channels,
                               int64_t height, int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                               const double *grad_y, int64_t out_channels, const float *rows, int32_t row_channels,
                               float *grad_features_t, float *h_partials, double *grad_x, void *stream);
/* the same with x as a float32 copy of the activations (every patch element is converted to float32 before the products
 * anyway, so the results are identical; half the bytes and registers of the gather).  Taken where the matrix-core kernel
 * runs the layer -- qiddm_qconv_train_x32_ok() returns 1 -- and QIDDM_ERR_UNSUPPORTED otherwise.                  */
int32_t qiddm_qconv_train_x32_ok(int64_t batch, int64_t in_channels, int64_t height, int64_t width, int64_t kh,
                                 int64_t kw, int64_t pad_h, int64_t pad_w, int64_t out_channels, int32_t row_channels);
int qiddm_qconv_train_backward_x32(int32_t n_qubits, const float *x, int64_t batch, int64_t in_channels,
                                   int64_t height, int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                                   const double *grad_y, int64_t out_channels, const float *rows,
                                   int32_t row_channels, float *grad_features_t, float *h_partials, double *grad_x,
                                   void *stream);
/* dL/dx without the feature-gradient matrix (qsim_qconv_dx.h): the fold commutes with the second product, so the thin-
 * product kernel only leaves 2 row_channels + 1 floats per output pixel (`pixel_rows`: qiddm_qconv_train_dx_elems()
 * floats) and a second kernel makes grad_x from them as a transposed convolution on the matrix cores -- C kh kw / (2
 * row_channels + 1) times less memory traffic than grad_features_t + fold.  Same-size convolutions (Ho = H, Wo = W) on the
 * matrix-core kernel with at most 32 input channels: qiddm_qconv_train_dx_elems() returns 0 for anything else, and
 * qiddm_qconv_train_backward_dx then QIDDM_ERR_UNSUPPORTED (use qiddm_qconv_train_backward).  Same h_partials; grad_x
 * agrees with the fold route to float32 rounding (the nine taps are summed in float32 instead of float64).
 * grad_y_batch_stride: elements between two images of grad_y, 0 = dense (out_channels Ho Wo); a channel slice of a wider
 * contiguous tensor -- one half of a concatenation's gradient -- is read in place.                                   */
int64_t qiddm_qconv_train_dx_elems(int32_t n_qubits, int64_t batch, int64_t in_channels, int64_t height, int64_t width,
                                   int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w, int64_t out_channels,
                                   int32_t row_channels);
int qiddm_qconv_train_backward_dx(int32_t n_qubits, const double *x, int64_t batch, int64_t in_channels,
                                  int64_t height, int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                                  const double *grad_y, int64_t grad_y_batch_stride, int64_t out_channels,
                                  const float *rows, int32_t row_channels, float *pixel_rows, float *h_partials,
                                  double *grad_x, void *stream);
/* The same backward for a convolution that is followed by a training-mode BatchNorm2d (every `net` of unet_simple,
 * reference nn/unet_simple.py:9-18): grad_out is dL/d(BatchNorm output), conv_y the convolution's own output and
 * bn_coef the (3, out_channels) coefficients of qiddm_batchnorm_backward_stats; the kernels form
 * dL/dy = coef[0] grad_out + coef[1] conv_y + coef[2] per channel while they load it, so the BatchNorm backward's
 * transform pass never runs.  With pixel_rows (qiddm_qconv_train_dx_elems() floats) dL/dx comes from the per-pixel
 * rows and grad_features_t may be NULL; otherwise as qiddm_qconv_train_backward.  qiddm_qconv_train_bn_ok() says whether
 * a layer has this form (1) or keeps the separate BatchNorm backward (0: the matrix-core kernel with 32 row channels has no
 * registers left for it; qiddm_qconv_train_backward_bn then returns QIDDM_ERR_UNSUPPORTED).                       */
int32_t qiddm_qconv_train_bn_ok(int64_t batch, int64_t in_channels, int64_t height, int64_t width, int64_t kh,
                                int64_t kw, int64_t pad_h, int64_t pad_w, int64_t out_channels, int32_t row_channels);
int qiddm_qconv_train_backward_bn(int32_t n_qubits, const double *x, int64_t batch, int64_t in_channels,
                                  int64_t height, int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                                  const double *grad_out, const double *conv_y, const double *bn_coef,
                                  int64_t out_channels, const float *rows, int32_t row_channels,
                                  float *grad_features_t, float *pixel_rows, float *h_partials, double *grad_x,
                                  void *stream);
int64_t qiddm_matrix_adjoint_partials(int64_t count);
int64_t qiddm_matrix_adjoint_workspace_bytes(const qiddm_circuit_t *circ, int64_t count);
int qiddm_matrix_adjoint(const qiddm_circuit_t *circ, const double *psi0, const double *lambda, int64_t count,
                         const void *gate_table, void *k_partials, void *workspace, int64_t workspace_bytes,
                         void *stream);

/* classical 1x1 convolution, float64 NCHW (the UNets' `final_conv`, reference nn/unet.py:160-166):
 * x (batch, in_channels, hw), weight (out_channels, in_channels), bias (out_channels) or NULL.          */
int qiddm_conv1x1_forward(const double *x, const double *weight, const double *bias, int64_t batch,
                          int64_t in_channels, int64_t out_channels, int64_t hw, double *y, void *stream);
/* backward of the ONE-output-channel head (torch autograd through `final_conv`, reference nn/unet.py:160-166, 177):
 * grad_x[b, c, p] = weight[c] grad_y[b, p], grad_weight[c] = sum grad_y x, grad_bias = sum grad_y -- one pass over x and
 * grad_y, per-workgroup partial sums (`partials`: qiddm_conv1x1_head_partials() x (in_channels + 1) float64) added in a
 * fixed order.  in_channels <= 32; grad_x / grad_weight / grad_bias may each be NULL.                            */
int64_t qiddm_conv1x1_head_partials(int64_t batch, int64_t hw);
int qiddm_conv1x1_head_backward(const double *x, const double *weight, const double *grad_y, int64_t batch,
                                int64_t in_channels, int64_t hw, double *grad_x, double *grad_weight,
                                double *grad_bias, double *partials, void *stream);

/* ---- density-matrix execution (hardware-noise study) ---------------------------------------------
 * Replaces PennyLane's `default.mixed` for the circuits the `*_noise.py` drivers run at sampling time
 * (src/mnist_noise.py:214-229; channels at nn/qdense.py:98-104, 255-261, 1410-1417).  The circuit is handed over
 * as a program of single-wire / two-wire ops (the caller expands templates and entangler rings); one workgroup
 * keeps one sample's rho (2^n x 2^n) in LDS or in `workspace`.  Forward only, n_qubits <= 8.
 *   QIDDM_MIX_ZERO            rho = |0..0><0..0|                      (a program starts with ZERO or AMP_EMBED)
 *   QIDDM_MIX_AMP_EMBED       AmplitudeEmbedding(features + enc_offset, pad_with, normalize)
 *   QIDDM_MIX_PHASE           RZ / PhaseShift on `wire`: angle = p + scale * angle_rows[a][sample] (a < 0: p only)
 *   QIDDM_MIX_RY              RY on `wire`, angle as above
 *   QIDDM_MIX_GATE            the 2x2 unitary gates[a] = (u00, u01, u10, u11) as (re, im) float64
 *   QIDDM_MIX_CZ / _CNOT      control `wire`, target `a`
 *   QIDDM_MIX_PHASE_DAMP / _AMP_DAMP / _DEPOL   PennyLane's PhaseDamping / AmplitudeDamping /
 *                             DepolarizingChannel with probability p on `wire`
 * program: HOST array (copied to the head of `workspace` on `stream`); angle_rows (n_rows, rows_ld >= batch),
 * features (batch, feat_ld), gates (n_gates, 8), out (batch, 2^n | n) float64 DEVICE arrays.            */
enum {
  QIDDM_MIX_ZERO = 0, QIDDM_MIX_AMP_EMBED, QIDDM_MIX_PHASE, QIDDM_MIX_RY, QIDDM_MIX_GATE, QIDDM_MIX_CZ,
  QIDDM_MIX_CNOT, QIDDM_MIX_PHASE_DAMP, QIDDM_MIX_AMP_DAMP, QIDDM_MIX_DEPOL
};
typedef struct qiddm_mixed_op {
  int32_t kind, wire, a, reserved;
  double p, scale;
} qiddm_mixed_op_t;
int64_t qiddm_mixed_workspace_bytes(int32_t n_qubits, int32_t dtype, int64_t batch, int32_t n_ops);
int qiddm_mixed_forward(int32_t n_qubits, int32_t dtype, const qiddm_mixed_op_t *program, int32_t n_ops,
                        const double *angle_rows, int64_t rows_ld, int32_t n_rows, const double *features,
                        int64_t feat_ld, int32_t n_features, double enc_offset, double pad_with,
                        const double *gates, int32_t n_gates, int32_t measure, int64_t batch, double *out,
                        int64_t out_ld, void *workspace, int64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QIDDM_HIP_H */
