/* l2o_abi.h -- C ABI of the MI355X-native Open-L2O inner-unroll library (libl2o_hip.so).
 *
 * Drop-in boundary for the ONE hot path of VITA-Group/Open-L2O that this project
 * accelerates: the model-free inner unroll loop of the coordinate-wise LSTM
 * optimizers (L2O-DM / L2O-RNNProp).  The reference has no FFI at all (it is
 * Python -> TensorFlow 1.14 graph ops); each entry point below cites the
 * reference interface (file:line under
 * "Model_Free_L2O/L2O-DM and L2O-RNNProp/", shorthand DM/) whose work it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer documented "device" is an fp32, contiguous, row-major HIP
 *     device pointer valid on the device of `stream`; the caller owns ALL memory
 *     (the library never allocates, frees or keeps device state between calls);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls
 *     are asynchronous on it and never synchronise the host;
 *   - return value: 0 = OK, L2O_ERR_ARG (-1) bad argument, L2O_ERR_UNSUPPORTED
 *     (-2) shape / configuration not implemented by the fused kernels (callers
 *     fall back to the step-granular entry points), L2O_ERR_HIP (-3) HIP runtime
 *     error, L2O_ERR_TIMEOUT (-4) a recoverable partner timeout reported by
 *     l2o_unroll_status; l2o_last_error() returns a thread-local message;
 *   - no C++ exception crosses the ABI; plain pointers and sizes only.
 */
#ifndef L2O_ABI_H_
#define L2O_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2O_ABI_VERSION 13

#define L2O_OK 0
#define L2O_ERR_ARG (-1)
#define L2O_ERR_UNSUPPORTED (-2)
#define L2O_ERR_HIP (-3)
#define L2O_ERR_TIMEOUT (-4)   /* ABI v12, l2o_unroll_status only: a workgroup of a kernel that exchanges data with a partner
                                  workgroup gave up waiting (bounded spin): that launch's outputs are invalid, the inputs
                                  the caller kept are not -- re-run it on an exchange-free form (L2O_OPT_PAIR = 0 /
                                  L2O_OPT_MLP_UNROLL = 0), which is what the Python host does */

/* optimizer-network kinds: networks.CoordinateWiseDeepLSTM (DM/networks.py:239-276),
 * networks.RNNprop (DM/networks.py:279-300) */
#define L2O_NET_CW 0
#define L2O_NET_RNNPROP 1

/* gradient preprocessing, StandardDeepLSTM.__init__/_build (DM/networks.py:180-188, 218-221):
 * tf.identity | preprocess.LogAndSign (DM/preprocess.py:42-70) | "fc" = ELU(Linear) */
#define L2O_PRE_IDENTITY 0
#define L2O_PRE_LOGSIGN 1
#define L2O_PRE_FC_ELU 2

/* optimizee kinds, the problems registry (DM/problems.py) */
#define L2O_PROB_SIMPLE 0      /* problems.simple / simple_multi_optimizer :41-70  f = sum x^2          */
#define L2O_PROB_QUADRATIC 1   /* problems.quadratic :73-101                                              */
#define L2O_PROB_LASSO 2       /* problems.lasso :103-134 and lasso_fixed :137-175                         */
#define L2O_PROB_RASTRIGIN 3   /* problems.rastrigin :177-213                                              */
#define L2O_PROB_SQUARE_COS 4  /* problems.square_cos :959-994                                             */
#define L2O_PROB_MLP 5         /* problems.mnist :254-288 (own entry point: l2o_mlp_fg)                    */

/* Hyper-parameters of one optimizer network: the `net_options` dict of
 * networks.factory (DM/networks.py:34-44) as used by util.get_config
 * (DM/util.py:99-109, 136-143, 251-263). */
typedef struct l2o_net_cfg {
  int32_t kind;         /* L2O_NET_*                                              */
  int32_t preprocess;   /* L2O_PRE_*                                              */
  int32_t n_layers;     /* len(layers): 0 (Linear only) or 2                      */
  int32_t hidden;       /* layers[i]; the fused kernels implement 20              */
  int32_t tanh_output;  /* StandardDeepLSTM(tanh_output=...) DM/networks.py:229   */
  int32_t reserved;
  double scale;         /* StandardDeepLSTM(scale=...) DM/networks.py:229-232     */
  double logsign_k;     /* LogAndSign(k) DM/preprocess.py:48                      */
  double beta1;         /* RNNProp MetaOptimizer(beta1, beta2) DM/meta_rnnprop_eval.py:230 */
  double beta2;         /* (python floats in the reference: kept as doubles so that the
                           fp32 constants are rounded exactly like TF rounds them)  */
  uint64_t options;     /* ABI v9: kernel A/B switches of THIS call, L2O_OPTW(option, value) words OR-ed together;
                           0 = every default (see "options" below).  The library keeps no option state.        */
} l2o_net_cfg;

/* One optimizee batch: the non-trainable variables that problems.<name>().build()
 * creates through tf.get_variable (DM/problems.py:84-96, 114-126, 149-165, 186-204),
 * sharded over GPUs by problem index.  The loss is a mean over the GLOBAL batch
 * (DM/problems.py:99, 131, 211), hence B_global. */
#define L2O_PROB_W_SHARED 1
#define L2O_PROB_FG_TWO_PASS 2   /* l2o_problem_fg / _hvp: the two-pass gradient kernel (default: one pass over the matrix) */

typedef struct l2o_problem {
  int32_t kind;          /* L2O_PROB_*                                                    */
  int32_t B_local;       /* problems held by this GPU                                     */
  int32_t B_global;      /* batch size in the reduce_mean (== B_local on one GPU)         */
  int32_t D;             /* optimizee parameters per problem (num_dims)                   */
  int32_t M;             /* rows of the matrix (== D except lasso_fixed)                  */
  int32_t flags;      /* L2O_PROB_W_SHARED: W is ONE [M, D] matrix used by every problem (batch stride 0) */
  double l1;             /* lasso `l`                                                     */
  double alpha;          /* rastrigin `alpha`                                             */
  const float* W;        /* device [B_local, M, D]  quadratic w / lasso w / rastrigin A   */
  const float* y;        /* device [B_local, M]     quadratic y / lasso y / rastrigin B   */
  const float* C;        /* device [B_local, D]: rastrigin C ; square_cos: the COLUMN SUMS of wcos
                            (sum_i (wcos c)_i = sum_j colsum_j c_j; alpha is fixed to 10); else NULL */
  const float* x_scale;  /* device [B_local, D] per-coordinate scale placeholder
                            (DM/meta_dm_train.py:336-338, 384) or NULL (== ones)          */
} l2o_problem;

/* ---- library info ------------------------------------------------------ */
int l2o_abi_version(void);
const char* l2o_last_error(void);
/* ABI v12.  l2o_build_id: 16 hex digits = sha256 over the sources the library was built from (csrc/Makefile passes it
 * in): measurements (rocprofv3 counter summaries under profiles/) carry it, so that bench.py can tell whether the
 * counters it quotes belong to the build it is timing.
 * l2o_last_unroll_form: which kernel the last l2o_unroll / l2o_unroll_record / l2o_unroll_reduce / l2o_mlp_unroll(_record)
 * call of THIS thread launched: L2O_FORM_* | (number of consecutive chunk launches << 8); 0 before the first call.
 * Diagnostic (thread-local, like l2o_last_error): the host labels its measurements with it and uses it to know whether a
 * launch exchanged data between workgroups (forms marked +: they can report L2O_ERR_TIMEOUT through l2o_unroll_status). */
const char* l2o_build_id(void);
int l2o_last_unroll_form(void);
#define L2O_FORM_UNROLL 1              /* k_unroll: one workgroup per problem, W in LDS                                  */
#define L2O_FORM_UNROLL_PAIR 2         /* + k_unroll_pair: every problem on two workgroups / CUs                         */
#define L2O_FORM_UNROLL_LDS 5          /* k_unroll_lds: one problem per CU, two waves per SIMD, fragments in LDS         */
#define L2O_FORM_UNROLL_CU 6           /* k_unroll_cu: the streaming form, four waves                                    */
#define L2O_FORM_UNROLL_CU8 7          /* k_unroll_cu8: the streaming form, eight waves                                  */
#define L2O_FORM_MLP_UNROLL 8          /* + k_mlp_unroll, fast instantiation, flat all-reduce                            */
#define L2O_FORM_MLP_UNROLL_HIER 9     /* + k_mlp_unroll, fast instantiation, XCD-hierarchical all-reduce                */
#define L2O_FORM_MLP_UNROLL_GENERIC 10 /* + k_mlp_unroll, generic loops                                                  */
#define L2O_FORM_MLP_XCD 11            /* + k_mlp_xcd: one optimizee instance per XCD (l2o_mlp_unroll_multi)             */

/* ---- options (ABI v9: per call, caller-owned) -------------------------------
 * A/B switches between kernels that compute the same thing (all results stay within the parity tolerance).  The
 * library keeps NO option state and never reads the environment: a call carries its switches in the caller's own
 * structs -- l2o_net_cfg.options = L2O_OPTW(option, value) | ..., l2o_problem.flags (L2O_PROB_FG_TWO_PASS),
 * l2o_mlp.flags (L2O_MLP_GENERIC) -- and a field left 0 means every default.  (ABI v6..v8 had a process-wide
 * l2o_set_option; removed.)  Apart from immutable hardware facts cached per device (CU count, measured co-resident
 * capacity) the library keeps no state between calls: per-launch state (status word, launch sequence) lives in the
 * caller-owned workspace. */
#define L2O_OPT_PAIR 0               /* 1*: l2o_unroll may split every problem over two CUs; 0: one CU per problem        */
#define L2O_OPT_PAIR_PLAIN_STORES 1  /* 1*: partners that CONFIRMED (XCC_ID handshake) they share an XCD publish their
                                        exchange granules with plain stores (L2-resident); 0: agent-scope stores always    */
#define L2O_OPT_UNROLL_CU 2          /* the streaming fused unroll for D > 128: 1*: on -- RNNProp's plain unroll on eight waves
                                        per workgroup with the fragments in LDS and the LSTM state in registers (k_unroll_cu8),
                                        everything else on four waves (k_unroll_cu) -- since ABI v12 the DM nets and the recording
                                        unrolls too, where their LDS image fits; 2: k_unroll_cu always; 3 / 4 / 5: k_unroll_cu8
                                        always (4 / 3 / 2 register tiles per wave); 0: such sizes run step-granular           */
#define L2O_OPT_FG_TWO_PASS 3        /* (l2o_problem.flags: L2O_PROB_FG_TWO_PASS)                                           */
#define L2O_OPT_MLP_GENERIC 4        /* (l2o_mlp.flags: L2O_MLP_GENERIC)                                                    */
#define L2O_OPT_BWD_BLOCKS 5         /* 0*: BPTT step kernels use one workgroup per CU; n > 0: n workgroups (L2O_OPTW_BWD_BLOCKS) */
#define L2O_OPT_BWD_KERNEL 6         /* 0*: matrix-core BPTT step (needs wpack); 1: fp32 tile kernel; 2: generic kernel     */
#define L2O_OPT_MLP_UNROLL 7         /* 1*: l2o_mlp_unroll available to the host layer (0: it reports "unsupported")       */
#define L2O_OPT_MLP_XCD_WAVES 8       /* l2o_mlp_unroll_multi (ABI v13): 0*: the form measured faster for the net (RNNProp: four
                                        waves per member, one per SIMD, eight tiles each stepped two at a time; the DM nets: eight
                                        waves, two per SIMD, four tiles each); 1: eight waves always; 2: four waves always.
                                        (Until ABI v11 option 8 was L2O_OPT_PAIR_NORMAL, removed in v12.)                         */
#define L2O_OPT_EXACT_GATES 9        /* 0*: LSTM gate GEMM as a 3-way bf16 split on v_mfma_f32_16x16x32_bf16 (fp32-level error,
                                        but the matrix pipe TRUNCATES small products inside an 8-slot group: a deterministic
                                        bias that shows as ~1e-5 drift at T = 1000); 1: v_mfma_f32_16x16x4_f32 (bit-equal to
                                        an fmaf chain) in the fused unroll kernels -- slower, for long-horizon evaluation    */
#define L2O_OPT_WPACK_NO_CLEAR 10     /* 0*: l2o_wpack_device clears the output buffer before it packs (the padding words of the
                                        fragment layout must be zero); 1: the caller vouches that this buffer already holds a
                                        pack of the same net configuration -- the padding is zero, the memset is skipped          */
#define L2O_OPT_MLP_HIER 11          /* 1*: l2o_mlp_unroll's fast form reduces the hidden pre-activations XCD-hierarchically (one
                                        fabric hop per step; falls back by itself when the workgroups are not placed round-robin
                                        over the XCDs); 0: the flat two-hop protocol */
#define L2O_OPT_ONE_LDS 12           /* large shards of the fused unroll (65 <= padded size <= 128) -- more problems than
                                        the #CU / 2 that one launch of the two-CU kernel holds: 0: consecutive chunk launches of
                                        that kernel (one workgroup per CU, fragments in registers); 1*: one problem per CU, two
                                        waves per SIMD, the gate-GEMM fragments in LDS (k_unroll_lds); 2: k_unroll_lds for every
                                        shard.  (3 was k_unroll_pair2, removed in ABI v12: it measured like 1)                  */
#define L2O_OPT_COUNT_ 13            /* (* = default) */
/* an option's 4-bit field in l2o_net_cfg.options: bit 3 = "set", bits 0-2 = the value.  Fields 0..11 sit at 4 * option;
 * bits 48-63 are the L2O_OPT_BWD_BLOCKS count, so option 12 uses the field that option 5 (that count) leaves unused */
#define L2O_OPT_FIELD_(o) ((o) == L2O_OPT_ONE_LDS ? L2O_OPT_BWD_BLOCKS : (o))
#define L2O_OPTW(o, v) ((uint64_t)(8u | ((unsigned)(v) & 7u)) << (4 * L2O_OPT_FIELD_(o)))
#define L2O_OPTW_BWD_BLOCKS(n) (((uint64_t)(n) & 0xffffu) << 48)

/* ---- co-residency (ABI v9) -------------------------------------------------
 * The two-CU unroll and l2o_mlp_unroll exchange data between workgroups that must be resident at the same time.  Before
 * every launch the library sizes them against the CUs this STREAM can use: the device's CU count (which already
 * reflects ROC_GLOBAL_CU_MASK and the compute-partition mode), the stream's own CU mask (hipExtStreamGetCUMask) and --
 * for restrictions HIP cannot see, e.g. HSA_CU_MASK -- the capacity MEASURED by this call: one probe launch of
 * one-workgroup-per-CU blocks that count each other; returns the number found co-resident (> 0, cached per device for
 * the life of the process) or a negative error.  Synchronizes `stream`.  scratch: >= 64 bytes of device memory.
 * A batch that does not fit runs as several smaller launches or on the one-CU kernel -- never as a launch that waits
 * for a partner that cannot be resident. */
int32_t l2o_coresident_workgroups(void* scratch, void* stream);

/* ---- weights: networks.factory / networks.save (DM/networks.py:34-62) ---
 * The `.l2l` dict {lstm_1:{w_gates,b_gates}, lstm_2:{...}, linear:{w,b},
 * input_projection:{w,b}} (Sonnet layouts: w_gates [in+H, 4H] gate order i,j,f,o;
 * Linear w [in,out]) is re-laid out on the HOST into MFMA-fragment order so that
 * every kernel loads each weight register with one coalesced dword load.
 * Pure host code: no device is touched. */
size_t l2o_wpack_floats(const l2o_net_cfg* cfg);
int l2o_wpack_host(const l2o_net_cfg* cfg,
                   const float* w_gates1, const float* b_gates1,   /* host [P+H,4H], [4H] */
                   const float* w_gates2, const float* b_gates2,   /* host [2H,4H], [4H]  */
                   const float* w_lin, const float* b_lin,         /* host [H or P,1], [1] */
                   const float* w_fc, const float* b_fc,           /* host [2,H],[H] (fc) or NULL */
                   float* wpack_out);                              /* host [l2o_wpack_floats] */

/* ---- LSTM state: net.initial_state_for_inputs (DM/networks.py:234-236, 273-276)
 * and the `update` assign of state_T (DM/meta.py:387-389).
 * Device state lives in a packed, tile-major layout (DESIGN.md "HBM layout"):
 * l2o_state_floats(B, D) floats for B problems of D coordinates.  Zero-filled
 * memory is the zero initial state.  pack/unpack convert from/to the reference
 * layout: per layer (hidden, cell) each [B*D, H] row-major. */
size_t l2o_state_floats(int64_t B, int64_t D);
int l2o_state_pack(const float* h1, const float* c1, const float* h2, const float* c2,
                   float* st, int64_t B, int64_t D, void* stream);
int l2o_state_unpack(const float* st, float* h1, float* c1, float* h2, float* c2,
                     int64_t B, int64_t D, void* stream);

/* ---- optimizee forward + gradient: build() + tf.gradients(fx, x)
 * (DM/meta.py:322, 344; closed forms DM/problems.py:98-99, 128-131, 206-211).
 * f_part[b] = per-problem loss term BEFORE the 1/B_global mean (so that
 * fx = sum_b f_part[b] / B_global); g = d fx / d x INCLUDING 1/B_global and the
 * x_scale chain rule.  g may be NULL (forward only). */
int l2o_problem_fg(const l2o_problem* prob, const float* x /* device [B_local,D] */,
                   float* f_part /* device [B_local] */, float* g /* device [B_local,D] or NULL */,
                   void* stream);

/* ---- Hessian-vector product of the analytic optimizees (ABI v6): out = (d g / d x) u with g what l2o_problem_fg
 * returns (1/B_global and the x_scale chain rule included), i.e. the term MetaOptimizer.meta_loss(...,
 * second_derivatives=True) adds to dL/dx_t when the optimizee gradient is NOT held constant (DM/meta.py:328-329):
 *   quadratic 2 s W^T W (s u) / B;  lasso s A^T A (s u) / B (the l1 term has no curvature);  rastrigin / square_cos
 *   additionally s (2 pi)^2 alpha C cos(2 pi x s) s u / B;  simple 2 s^2 u.
 * scratch: device [B_local] floats. */
int l2o_problem_hvp(const l2o_problem* prob, const float* x /* device [B_local,D] */, const float* u /* [B_local,D] */,
                    float* out /* device [B_local,D] */, float* scratch, void* stream);

/* ---- neural optimizee: problems.mnist (DM/problems.py:246-288) = mean sparse-softmax
 * cross-entropy of snt.nets.MLP([n_hidden, n_out]) on a minibatch gathered from a resident
 * dataset; forward + tf.gradients w.r.t. the four variables (DM/meta.py:322, 344).
 * One hidden layer (util.get_config("mnist"): layers=(20,), DM/util.py:147-149).
 * loss[0] = the scalar loss; gradients may all be NULL (forward only). */
#define L2O_MLP_GENERIC 1   /* l2o_mlp_fg: the generic-width kernels also for hidden width 20 */
typedef struct l2o_mlp {
  int32_t n_in;          /* 784                                                      */
  int32_t n_hidden;      /* layers[0] (<= 32)                                        */
  int32_t n_out;         /* 10 (<= 16)                                               */
  int32_t batch;         /* minibatch size (<= 256)                                  */
  int32_t activation;    /* 0 = sigmoid, 1 = relu  (DM/problems.py:260-265)          */
  int32_t n_data;        /* rows of `images`                                         */
  int32_t flags;         /* ABI v9: L2O_MLP_GENERIC                                  */
  int32_t reserved;
  const float* images;   /* device [n_data, n_in]                                    */
  const int32_t* labels; /* device [n_data]                                          */
} l2o_mlp;
size_t l2o_mlp_scratch_floats(const l2o_mlp* mlp);
int l2o_mlp_fg(const l2o_mlp* mlp, const int32_t* indices /* device [batch] rows of the minibatch */,
               const float* w1 /* device [n_in,n_hidden] */, const float* b1 /* [n_hidden] */,
               const float* w2 /* device [n_hidden,n_out] */, const float* b2 /* [n_out] */,
               float* loss /* device [1] */, float* gw1, float* gb1, float* gw2, float* gb2,
               float* scratch /* device [l2o_mlp_scratch_floats] */, void* stream);

/* problems.mnist with MORE than one hidden layer (ABI v13; DM/problems.py:254-288 with layers = (20, 20): DM/util.py:157-163
 * "mnist_deeper").  The step-granular evaluation only -- loss + gradients of one minibatch, two launches; the fused unrolls
 * serve the one-hidden-layer optimizee.  w / g: HOST arrays of 2 (n_hidden_layers + 1) device pointers in Sonnet's order
 * linear_0/w [n_in, h0], linear_0/b [h0], linear_1/w [h0, h1], ..., linear_L/w [h_{L-1}, n_out], linear_L/b [n_out]
 * (g may be NULL: forward only).  n_hidden_layers in [1, 3], widths <= 32, n_out <= 16, n_in <= 1024, batch <= 4096. */
typedef struct l2o_mlp_deep {
  int32_t n_in, n_out, batch, activation /* 0 sigmoid, 1 relu */, n_data, n_hidden_layers;
  int32_t hidden[3];
  int32_t reserved;
  const float* images;   /* device [n_data, n_in] */
  const int32_t* labels; /* device [n_data]       */
} l2o_mlp_deep;
size_t l2o_mlp_deep_scratch_floats(const l2o_mlp_deep* mlp);
int l2o_mlp_deep_fg(const l2o_mlp_deep* mlp, const int32_t* indices /* device [batch] */, const float* const* w,
                    float* loss /* device [1] */, float* const* g, float* scratch /* device */, void* stream);

/* ---- the fused unroll for the neural optimizee (ABI v6): MetaOptimizer.meta_loss's tf.while_loop
 * (DM/meta.py:338-376; RNNProp DM/meta_rnnprop_eval.py time_step) over problems.mnist (DM/problems.py:246-288) as ONE
 * persistent launch: T x { fx_t = loss(minibatch_t; x_t * s); g = s * grad; delta, state = net(g, state) for each of
 * the four variables (one shared net, DM/meta.py:330-336); x += delta } then fx_T on minibatch_T.
 *   indices   device int32 [T + 1][batch]: the minibatch of every evaluation (DM/problems.py:282-286 draws one per
 *             evaluation of the loss; the host draws them up front in the reference's order)
 *   x, st, m, v, x_scale   HOST arrays of 4 device pointers in the order w1 [n_in, H], b1 [H], w2 [H, O], b2 [O]
 *             (st: packed state of l2o_state_floats(1, n) floats; m, v: RNNProp only; x_scale may be NULL or
 *             hold NULLs); all updated in place == the harness' `update` op
 *   fx        device [T + 1]: the scalar loss of every evaluation
 *   workspace caller-owned device scratch of l2o_mlp_unroll_workspace_bytes() bytes, zeroed once by the caller;
 *             header as for l2o_unroll (sticky status word -> l2o_unroll_status, launch sequence word)
 * One workgroup per 64 coordinates, all co-resident (n_tiles / 4 <= #CUs): l2o_mlp_unroll_supported() == 0
 * otherwise (and for n_in * H % 64 != 0, H outside [8, 32], O > 16, batch > 128) -- callers then run
 * l2o_mlp_fg + l2o_cwlstm_step_multi per step.  Returns 2 (since ABI v10) when the FAST instantiation applies (the
 * reference's shape: hidden 20, 10 classes, minibatch 64: 11 us per step against 22 us of the step-granular launches),
 * 1 for the generic loops (at minibatch 128 no faster than the step-granular path: 39 vs 38 us per step). */
int l2o_mlp_unroll_supported(const l2o_net_cfg* cfg, const l2o_mlp* mlp, void* stream);
size_t l2o_mlp_unroll_workspace_bytes(const l2o_mlp* mlp);
int l2o_mlp_unroll(const l2o_net_cfg* cfg, const float* wpack /* device */, const l2o_mlp* mlp,
                   const int32_t* indices, float* const* x, float* const* st, float* const* m, float* const* v,
                   const float* const* x_scale, int32_t T, int32_t step0, float* fx, void* workspace, void* stream);
/* The same unroll, also recording what the meta-gradient needs (ABI v10; the MLP counterpart of l2o_unroll_record):
 * T optimizer steps and T + 1 gradient evaluations (indices must hold T + 1 rows: the one at x_T is the g_final of
 * l2o_cwlstm_bwd_unroll).  Per variable k (w1, b1, w2, b2; n_k coordinates):
 *   hist.st[k]  [T][l2o_state_floats(1, n_k)]  the packed LSTM state BEFORE step t
 *   hist.g[k]   [T + 1][n_k]                   the gradient at x_t (times x_scale), slot T = at x_T
 *   hist.m[k], hist.v[k]  [T + 1][n_k]         RNNProp: the moments AFTER step t in slot t + 1 (slot 0 is not written);
 *                                              NULL for the DM nets */
typedef struct l2o_mlp_hist {
  float* st[4];
  float* g[4];
  float* m[4];
  float* v[4];
} l2o_mlp_hist;
int l2o_mlp_unroll_record(const l2o_net_cfg* cfg, const float* wpack /* device */, const l2o_mlp* mlp,
                          const int32_t* indices, float* const* x, float* const* st, float* const* m, float* const* v,
                          const float* const* x_scale, int32_t T, int32_t step0, float* fx, const l2o_mlp_hist* hist,
                          void* workspace, void* stream);

/* The same unroll for UP TO EIGHT independent optimizee instances in ONE launch, every instance confined to one XCD
 * (ABI v13, csrc/l2o_mlp_xcd.h): the replicas of BASELINE config 5 / a meta-training batch of optimizees
 * (DM/meta_rnnprop_train.py:397-423 runs one such unroll per sess.run; N of them are N independent unrolls).  Instance j
 * runs on the 32 CUs of XCD j -- eight tile-steps per SIMD and step, the all-reduce of the hidden pre-activations through
 * that XCD's own L2 -- so a launch of 8 instances has ~3-4 x the throughput of 8 l2o_mlp_unroll launches, and a launch of
 * ONE instance a longer latency than l2o_mlp_unroll: the caller chooses per call.
 *   inst[j]   indices / x / st / m / v / x_scale / fx of instance j, each as in l2o_mlp_unroll (all instances share the
 *             network, `mlp` (shape and data set) and T / step0; every instance draws its own minibatches)
 * Supported (l2o_mlp_unroll_multi_supported() != 0) for the reference's shape only (hidden 20, 10 classes, minibatch 64,
 * n_in * 20 + 230 coordinates <= 16 368), 1 <= n_inst <= 8, and a device whose 8 XCDs x 32 CUs are all available to the
 * stream.  A team that does not assemble (masked / shared device) raises the sticky status word -> L2O_ERR_TIMEOUT. */
typedef struct l2o_mlp_instance {
  const int32_t* indices;      /* device [T + 1][batch] */
  float* x[4];
  float* st[4];
  float* m[4];                 /* RNNProp only (else NULL) */
  float* v[4];
  const float* x_scale[4];     /* NULLs allowed */
  float* fx;                   /* device [T + 1] */
} l2o_mlp_instance;
int l2o_mlp_unroll_multi_supported(const l2o_net_cfg* cfg, const l2o_mlp* mlp, int32_t n_inst, void* stream);
size_t l2o_mlp_unroll_multi_workspace_bytes(const l2o_mlp* mlp, int32_t n_inst);
int l2o_mlp_unroll_multi(const l2o_net_cfg* cfg, const float* wpack /* device */, const l2o_mlp* mlp,
                         const l2o_mlp_instance* inst /* host [n_inst] */, int32_t n_inst, int32_t T, int32_t step0,
                         void* workspace, void* stream);

/* ---- one optimizer step on a gradient panel: the closure `update`
 * (DM/meta.py:319-336; RNNProp DM/meta_rnnprop_train.py:371-395) for ONE variable:
 * preprocess -> 2-layer coordinate-wise LSTM -> Linear -> (tanh) * scale -> x += delta
 * (DM/networks.py:207-232, 254-271, 287-295; DM/meta.py:353).
 * RNNProp: m, v are updated in place and the inputs (m~, g~) are formed with the
 * bias-correction powers beta^k, k = step + t (DM/meta_rnnprop_train.py:383-388);
 * pass pow1 = beta1^k, pow2 = beta2^k.  m, v are ignored (may be NULL) for L2O_NET_CW. */
int l2o_cwlstm_step(const l2o_net_cfg* cfg, const float* wpack /* device */,
                    const float* g /* device [B,D] */, float* m, float* v /* device [B,D] */,
                    double pow1, double pow2,
                    float* st /* device, packed, in-out */, float* x /* device [B,D] in-out */,
                    int64_t B, int64_t D, void* stream);

/* ---- the same step for ANY `layers` tuple (ABI v6): StandardDeepLSTM(layers=...) builds one snt.LSTM per entry
 * (DM/networks.py:192-200); the matrix-core kernels above implement the harness' (20, 20), this entry point every
 * other stack of 1..3 layers with hidden sizes <= 64 (the reference's own tests use layers=(1,) and (1, 1),
 * L2O-Swarm/src/networks_test.py:33-47), one thread per coordinate, plain fp32.  Weights stay in their Sonnet
 * layouts (device pointers).  State: l2o_gen_state_floats(net, N) floats -- per layer hidden [N][H_l] then cell
 * [N][H_l]; zero memory is the zero state.
 * direct_inputs = 1 (L2O_NET_RNNPROP only): the plugin contract of RNNprop._build, `net(m, g, prev_state)`
 * (DM/networks.py:287-295): g holds g~, m_tilde holds m~, no moments are read or written. */
typedef struct l2o_gen_net {
  int32_t n_layers;            /* len(layers): 1..3                                           */
  int32_t hidden[3];           /* layers[l] <= 64                                             */
  int32_t in_dim;              /* width after preprocessing: 1 identity, 2 LogAndSign, fc dim */
  int32_t direct_inputs;
  const float* w_gates[3];     /* device [in_l + H_l, 4 H_l], gate order i, j, f, o           */
  const float* b_gates[3];     /* device [4 H_l]                                              */
  const float* w_lin;          /* device [H_last, 1]                                          */
  const float* b_lin;          /* device [1]                                                  */
  const float* w_fc;           /* device [2, in_dim] (fc) or NULL                             */
  const float* b_fc;           /* device [in_dim]                                             */
} l2o_gen_net;
size_t l2o_gen_state_floats(const l2o_gen_net* net, int64_t N);
int l2o_cwlstm_step_generic(const l2o_net_cfg* cfg /* kind, preprocess, tanh_output, scale, logsign_k, betas */,
                            const l2o_gen_net* net, const float* g /* device [N] */,
                            const float* m_tilde /* device [N], direct_inputs only */,
                            float* m, float* v /* device [N], RNNProp without direct_inputs */,
                            double pow1, double pow2, float* state /* device, in-out */,
                            float* x /* device [N] in-out: x += delta */, int64_t N, void* stream);

/* One step of back-propagation through time for the same ANY-`layers` stack (ABI v9): what
 * tf.train.AdamOptimizer(lr).minimize(loss) of meta_minimize (DM/meta.py:398-414) differentiates through `update`
 * (DM/meta.py:319-336) when networks.factory built a stack other than (20, 20) (DM/networks.py:157 accepts any tuple).
 * The step is recomputed from st_prev; emitted per layer l: act[l] [N][in_l + H_l] = [input_l | h_l(t-1)] and
 * dz[l] [N][4 H_l] = dL/d(gate pre-activations) -- the weight gradients of the step are act[l]^T dz[l] (biases: the
 * column sums of dz[l]), h_last^T dd for the output Linear and, RNNProp, feats^T du for the input projection.
 * carry_in / carry_out: dL/dh_l, dL/dc_l per layer in the l2o_gen_state_floats layout (zeros into the last step).
 * m, v: the moments AFTER the step's update.  dg (identity / LogAndSign, optional): dL/dg_t for second_derivatives. */
typedef struct l2o_gen_bwd_io {
  const float* g;          /* device [N]                                  */
  const float* m;          /* device [N], RNNProp                         */
  const float* v;
  const float* st_prev;    /* device, l2o_gen_state_floats: state before the step */
  const float* dx_next;    /* device [N]: dL/d(delta_t)                   */
  const float* carry_in;   /* device, state layout                        */
  float* carry_out;
  float* act[3];
  float* dz[3];
  float* tc;               /* device scratch [N * sum_l H_l]              */
  float* h_last;           /* device [N][H_last]                          */
  float* dd;               /* device [N]                                  */
  float* feats;            /* device [N][2], RNNProp                      */
  float* du;               /* device [N][in_dim], RNNProp                 */
  float* dg;               /* device [N] or NULL                          */
} l2o_gen_bwd_io;
int l2o_cwlstm_bwd_step_generic(const l2o_net_cfg* cfg, const l2o_gen_net* net, const l2o_gen_bwd_io* io, double pow1,
                                double pow2, int64_t N, void* stream);

/* The same update for several variables that share one network in ONE launch (the reference
 * applies `net` to every variable of a subset inside the same time step, DM/meta.py:330-336;
 * problems.mnist has four: mlp/linear_{0,1}/{w,b}).  `segs` is a HOST array of 1..8 panels. */
typedef struct l2o_step_seg {
  const float* g;      /* device [B,D] */
  float* m;            /* device [B,D], RNNProp only */
  float* v;
  float* st;           /* device, packed, in-out */
  float* x;            /* device [B,D] in-out */
  int64_t B, D;
  float* st_out;       /* NULL: in place.  Else the new state / moments are written here and st / m / v stay */
  float* m_out;        /* as they were: a caller that records the unroll for the meta-gradient chains its     */
  float* v_out;        /* history buffers through the steps instead of copying them (DM/meta.py:398-414)     */
} l2o_step_seg;
int l2o_cwlstm_step_multi(const l2o_net_cfg* cfg, const float* wpack /* device */, const l2o_step_seg* segs,
                          int32_t nseg, double pow1, double pow2, void* stream);

/* ---- meta-gradient: one step of back-propagation-through-time of the optimizer network,
 * i.e. what tf.train.AdamOptimizer(lr).minimize(loss) differentiates in
 * MetaOptimizer.meta_minimize (DM/meta.py:398-414) with the optimizee gradient held constant
 * (tf.stop_gradient, DM/meta.py:328-329).  Called for t = T-1 .. 0 with
 *   dx_next  = dL/dx_{t+1} = sum_{tau > t} g_tau   (loss = sum_t f(x_t), DM/meta.py:376)
 *   carry_in = (dh1, dc1, dh2, dc2) from step t+1 ([4][N][H], zeros for the last step)
 * it recomputes the step's forward from the state saved BEFORE the step (`st_prev`, packed)
 * and writes the rows from which the host forms the weight gradients as GEMMs over
 * (steps x coordinates):  dW1 = act1^T dz1, db1 = sum dz1, dW2 = act2^T dz2, db2 = sum dz2,
 * dw_lin = h2^T dd, db_lin = sum dd, (RNNProp) dW_fc = feats^T du, db_fc = sum du.
 * Gate column order of dz* is Sonnet's (i, j, f, o).  layers=(): only act1 [N,2] and dd. */
typedef struct l2o_net_weights {      /* device pointers, Sonnet / .l2l layouts */
  const float *w_gates1, *b_gates1, *w_gates2, *b_gates2, *w_lin, *b_lin, *w_fc, *b_fc;
  const float* wpack;     /* optional: device copy of l2o_wpack_host's output for the SAME weights.  With it the
                           * tile-aligned BPTT step runs on the matrix cores (k_cwlstm_bwd_mfma); NULL keeps the
                           * fp32 kernels.  Results agree to fp32 rounding. */
} l2o_net_weights;
typedef struct l2o_bwd_io {
  const float* g;          /* device [N]   gradient fed to the net at this step               */
  const float* m;          /* device [N]   RNNProp moments AFTER this step's update (or NULL) */
  const float* v;
  const float* st_prev;    /* device packed LSTM state BEFORE the step (NULL for layers=())   */
  const float* dx_next;    /* device [N]                                                      */
  const float* carry_in;   /* device [4][N][H] (NULL for layers=())                           */
  float* carry_out;        /* device [4][N][H]                                                */
  float* act1;             /* device [N][P+H]  (P = 1 | 2 | H for identity | LogAndSign | fc); layers=(): [N][2] */
  float* dz1;              /* device [N][4H]                                                  */
  float* act2;             /* device [N][2H]                                                  */
  float* dz2;              /* device [N][4H]                                                  */
  float* h2;               /* device [N][H]                                                   */
  float* dd;               /* device [N]   dL/d(output Linear)                                */
  float* feats;            /* device [N][2]  RNNProp (m~, g~)                                 */
  float* du;               /* device [N][H]  RNNProp d/d(fc pre-activation)                   */
  /* Row strides in floats of the two groups of emitted rows, or 0 for the dense layouts above.
   * With a_stride / b_stride the caller interleaves  A = [act1 | act2 | h2 | feats | 1]  and
   * Bm = [dz1 | dz2 | dd | du]  in two row-major matrices (the pointers above are then column
   * offsets into them) and obtains EVERY weight gradient from the single product A^T Bm. */
  int64_t a_stride;        /* act1, act2, h2, feats */
  int64_t b_stride;        /* dz1, dz2, dd, du      */
  float* dg;               /* optional, device [N]: dL/d(the gradient fed to the net at this step), through the
                            * preprocessing -- what second_derivatives=True back-propagates into the optimizee
                            * (DM/meta.py:328-329 skips the stop_gradient); DM nets (identity / LogAndSign) only */
} l2o_bwd_io;
int l2o_cwlstm_bwd_step(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_io* io,
                        double pow1, double pow2, int64_t B, int64_t D, void* stream);

/* The same step for up to 8 panels (variables) that share the network, in ONE launch
 * (the variables of a subset, DM/meta.py:330-336; problems.mnist has four).  Panel s occupies
 * the tiles [T_{s-1}, T_s), T_s = sum_{j<=s} ceil(B_j D_j / 16), and the rows 16 * tile of the
 * shared matrices  A [rows][KA] = [act1 | act2 | h2 | feats | 1],  Bm [rows][KB] = [dz1 | dz2 | dd | du]
 * (rows = 16 * T_last; the rows of a ragged last tile that do not exist are left untouched) and
 * of the carries [4][rows][H].  Every panel needs D % 16 == 0 or B == 1. */
typedef struct l2o_bwd_seg {
  const float* g;          /* device [B*D] */
  const float* m;          /* RNNProp moments AFTER the step (or NULL) */
  const float* v;
  const float* st_prev;    /* device packed LSTM state BEFORE the step */
  const float* dx_next;    /* device [B*D] */
  int64_t B, D;
} l2o_bwd_seg;
int l2o_cwlstm_bwd_multi(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_seg* segs, int32_t nseg,
                         const float* carry_in, float* carry_out, float* A, float* Bm,
                         double pow1, double pow2, void* stream);

/* ALL T steps of the back-propagation through a recorded unroll in ONE launch (the whole of what
 * tf.gradients walks back through the while_loop of DM/meta.py:361-368): a wave keeps its tile's
 * (dh, dc) carries in registers from step T-1 down to step 0, so only the recorded history comes in
 * and the rows of A / Bm go out.  Needs l2o_net_weights.wpack (matrix-core kernel).  Any D (since ABI v10): the tiles
 * of a panel are PER PROBLEM, ceil(D / 16) each with a ragged last one -- the packed-state layout the forward kernels
 * record -- so panel s occupies B_s ceil(D_s / 16) tiles and 16 times as many rows (for D % 16 == 0 or B == 1 that is
 * the ceil(B D / 16) of l2o_cwlstm_bwd_multi).
 *   table     device array [T][nseg][5] of device pointers: g, m, v, st_prev, dx_next of panel s at
 *             step t (same meaning as l2o_bwd_seg; m, v NULL for the DM nets).  dx_next may be NULL:
 *             then dL/d(delta_t) = g_final + sum_{tau > t} g_tau, the gradient of loss = sum_t fx_t
 *             (DM/meta.py:376) through x_{t+1} = x_t + delta_t, accumulated in a register.
 *   A, Bm     [T][rows][KA], [T][rows][KB] (rows as in l2o_cwlstm_bwd_multi); EVERY row is written -- the rows
 *             of a ragged last tile that do not exist as zeros, so the caller need not clear the buffers
 *   carry_in  [4][rows][H] gradient w.r.t. the state after step T-1, or NULL (zeros);
 *   carry_out gradient w.r.t. the state before step 0, or NULL (not wanted)
 *   step0     RNNProp: step t uses the bias corrections 1 - beta^(step0 + t) (DM/util.py:59-60) */
typedef struct l2o_bwd_unroll_seg {
  int64_t B, D;
  const float* g_final;    /* device [B*D] gradient at x_T, or NULL when the table carries dx_next */
} l2o_bwd_unroll_seg;
int l2o_cwlstm_bwd_unroll(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_unroll_seg* segs,
                          int32_t nseg, const float* const* table, int32_t T, int64_t step0,
                          const float* carry_in, float* carry_out, float* A, float* Bm, void* stream);
/* The same launch with the A rows WITHOUT their duplicated columns (ABI v12).  A row of step t is
 * [in | h1(t-1)] [h1(t) | h2(t-1)] [h2(t)] [feats] [1], and h1(t-1), h2(t-1) are the h1, h2 columns of step t - 1's row: 40 of
 * the 82 (83, 103) columns are written twice and read back by the contraction.  Here a row keeps [in | h1(t) | h2(t) |
 * feats | 1] (KA - 40 floats) and Ac holds T + 1 blocks of `rows` rows: block t + 1 = step t, block 0 = the state before
 * step 0 (zeros in the other columns).  17 % fewer bytes out of this kernel and into l2o_cwlstm_wgrad_compact, which
 * forms the same [KA][KB] result from the two blocks a step's operands live in.  (h1(t-1) of the product is then the
 * value this kernel RECOMPUTED for step t - 1, not the recorded one: equal up to the forward kernels' fp32 rounding.)
 * Matrix-core BPTT kernel only (L2O_OPT_BWD_KERNEL = 0); otherwise as l2o_cwlstm_bwd_unroll. */
int l2o_cwlstm_bwd_unroll_compact(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_unroll_seg* segs,
                                  int32_t nseg, const float* const* table, int32_t T, int64_t step0,
                                  const float* carry_in, float* carry_out, float* Ac /* [T+1][rows][KA-40] */,
                                  float* Bm /* [T][rows][KB] */, void* stream);

/* ---- the weight-gradient contraction (ABI v6): out [KA][KB] = A^T B for A [R][KA], B [R][KB] dense row-major device
 * matrices (KA <= 112, KB <= 192, any R): with A = [act1 | act2 | h2 | feats | 1] and B = [dz1 | dz2 | dd | du] as
 * written by l2o_cwlstm_bwd_unroll / _multi / _step, every weight gradient of the unroll is a block of `out`
 * (what tf.gradients accumulates over the while_loop, DM/meta.py:398-414).  fp32 products and sums
 * (v_mfma_f32_16x16x4_f32), split over the rows with a fixed-order reduction: bit-reproducible.
 * workspace: l2o_atb_workspace_bytes() bytes of device scratch. */
size_t l2o_atb_workspace_bytes(int64_t R, int32_t KA, int32_t KB);
int l2o_atb(const float* A, const float* B, int64_t R, int32_t KA, int32_t KB, float* out, void* workspace, void* stream);
/* The same contraction restricted to what IS a weight gradient (ABI v8): G [KA][KB] = A^T Bm for the rows written by
 * l2o_cwlstm_bwd_unroll / _multi / _step of a layers=(20,20) net (KA, KB from l2o_cwlstm_wgrad_dims), computing only
 * the blocks  [in | h1_prev]^T dz1,  [h1 | h2_prev]^T dz2,  h2^T dd,  (RNNProp) feats^T du  and the bias row
 * 1^T [dz1 | dz2 | dd | du]  -- 38 of the 66 (43 of 84) 16 x 16 tiles; every other entry of G is written as 0.
 * Arithmetic: by default the products run on the bf16 matrix pipe as a 3-way split of both operands (6 bf16 MFMAs per
 * tile and 32-row block, fp32 accumulation; error against a float64 product ~2e-8 of sum |a||b|, the same as the fp32
 * pipe's); with L2O_OPT_EXACT_GATES in cfg->options the fp32 pipe of l2o_atb (bit-equal to it on those blocks).  Both
 * are bit-reproducible run to run.  Same workspace (l2o_atb_workspace_bytes(R, KA, KB)). */
int32_t l2o_cwlstm_wgrad_dims(const l2o_net_cfg* cfg, int32_t* KA, int32_t* KB);
int l2o_cwlstm_wgrad(const l2o_net_cfg* cfg, const float* A, const float* Bm, int64_t R, float* G /* device [KA][KB] */,
                     void* workspace, void* stream);
/* ... from the compact rows of l2o_cwlstm_bwd_unroll_compact (ABI v12): Ac [T + 1][rows][KA - 40], Bm [T][rows][KB]; G and
 * the workspace (l2o_atb_workspace_bytes(T * rows, KA, KB)) as above.  bf16 pipe only: L2O_ERR_UNSUPPORTED with
 * L2O_OPT_EXACT_GATES (callers then use the plain pair). */
int l2o_cwlstm_wgrad_compact(const l2o_net_cfg* cfg, const float* Ac, const float* Bm, int32_t T, int64_t rows,
                             float* G /* device [KA][KB] */, void* workspace, void* stream);

/* ---- the meta-step on the device (ABI v5): tf.train.AdamOptimizer(learning_rate).minimize(loss)
 * (DM/meta.py:410-414) without a host round trip of the weights.
 * l2o_adam_step: TF 1.x `_apply_dense` on one flat fp32 vector (all device pointers, n elements):
 *   m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;  w <- w - lr_t m / (sqrt(v) + eps),
 *   lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) computed by the caller; every operation rounded
 *   separately (bit-equal to the NumPy expression of the host path).
 * l2o_wpack_device: l2o_wpack_host for weights that live on the device (w: Sonnet layouts, `wpack`
 *   member ignored); writes l2o_wpack_floats(cfg) floats, bit-equal to the host packer's output. */
int l2o_adam_step(float* w, float* m, float* v, const float* g, int64_t n, float lr_t, double beta1,
                  double beta2, double epsilon, void* stream);   /* fp32(beta), fp32(1 - beta), fp32(eps) are used */
/* The same update, conditional ON THE DEVICE on the status word of the unroll the gradients come from (ABI v10):
 * unroll_workspace = the workspace of that l2o_unroll_record call (device pointer; NULL = unconditional).  If the
 * kernel reported a partner timeout (status != 0: its recorded history is garbage), w, m and v are left untouched.
 * Lets a caller enqueue the meta-step BEHIND the unroll and its back-propagation without waiting for the status on
 * the host first; it still reads the status after its sync (l2o_unroll_status) and then knows the update did not run. */
int l2o_adam_step_guarded(float* w, float* m, float* v, const float* g, int64_t n, float lr_t, double beta1,
                          double beta2, double epsilon, const void* unroll_workspace, void* stream);
/* The same update with the gradient GATHERED (ABI v10): g_i = map[i] >= 0 ? G[map[i]] : 0, map a device int32 [n].
 * With G = the [KA][KB] result of l2o_cwlstm_wgrad and map = the position of every weight of the flat Sonnet-layout
 * buffer inside its gradient block, the meta-step reads the contraction's output in place (no slicing / concatenation
 * launches in between).  unroll_workspace: as for l2o_adam_step_guarded (NULL = unconditional). */
int l2o_adam_step_gather(float* w, float* m, float* v, const float* G, const int32_t* map, int64_t n, float lr_t,
                         double beta1, double beta2, double epsilon, const void* unroll_workspace, void* stream);
int l2o_wpack_device(const l2o_net_cfg* cfg, const l2o_net_weights* w, float* wpack_out /* device */,
                     void* stream);

/* ---- the fused unroll: MetaOptimizer.meta_loss's tf.while_loop
 * (DM/meta.py:338-376; RNNProp DM/meta_rnnprop_eval.py time_step) as ONE persistent
 * launch: T x { fx_t = f(x_t*s); g = s*grad f; delta,state = net(g,state); x += delta }
 * then fx_T.  x, st (and m, v) are updated in place == the harness' `update` op.
 * fx_part[t*B_local + b] receives problem b's loss term at step t (t = 0..T).
 * step0 = the harness-fed `step` (DM/util.py:59-60, 85-86); RNNProp only.
 * Returns L2O_ERR_UNSUPPORTED when (problem size, net) has no fused kernel.
 *
 * workspace: caller-owned device scratch of l2o_unroll_workspace_bytes() bytes, or NULL.
 * With a workspace every problem is split over TWO workgroups (two CUs) that exchange their partial
 * residuals once per step through tagged 8-byte granules in the
 * workspace; a launch holds at most (CUs usable by the stream) / 2 problems -- both halves of each
 * co-resident, see l2o_coresident_workgroups -- and a larger shard runs as consecutive launches of
 * equal chunks.  Without a workspace: one workgroup per problem.
 * The first 4 bytes of the workspace are a STICKY status word the kernel raises if a partner
 * never showed up (bounded spin, no hang; the results of that launch are then invalid): after
 * synchronising, copy them to the host and pass them to l2o_unroll_status(); the caller clears
 * the word (writes 0) once it has handled the error.  The next 4 bytes are a launch sequence
 * number the kernels maintain (the salt of the exchange tags).  Bytes 8..11 are a FAULT-INJECTION word for
 * tests (ABI v12; 0 in production and after l2o_unroll_workspace_init): while it is non-zero every launch of
 * a kernel that exchanges data behaves as if its partners never showed up -- it raises the status at once
 * and its results are invalid; the exchange-free forms ignore it.  Bytes 16..23 (ABI v12, int64): the shader-clock
 * cycles (s_memtime) wave 0 of workgroup 0 spent in the step loop of the LAST launch on this workspace (the two-CU
 * kernel, k_unroll_lds, l2o_mlp_unroll) -- T steps + the final loss evaluation --, bytes 24..31 the same wave's cycles
 * from kernel entry to its last store: measurements in cycles that need no clock-frequency assumption (bench.py's
 * roofline block).  A workspace must start zeroed
 * (see l2o_unroll_workspace_init / _layout below) and must not be shared by two streams at the same time.
 * Problems beyond the LDS-resident sizes (D <= 512, D % 4 == 0, any M) run the streaming
 * form: one workgroup per problem, the matrix streamed once per step, x / state / moments on-chip
 * for the whole unroll; it needs no workspace (l2o_unroll_workspace_bytes() == 0). */
size_t l2o_unroll_workspace_bytes(const l2o_net_cfg* cfg, const l2o_problem* prob, int32_t T);
int l2o_unroll(const l2o_net_cfg* cfg, const float* wpack /* device */,
               const l2o_problem* prob, float* x /* device [B_local,D] in-out */,
               float* st /* device, packed, in-out */, float* m, float* v,
               int32_t T, int32_t step0,
               float* fx_part /* device [(T+1)*B_local] */, void* workspace, void* stream);
/* The same unroll that also RECORDS what back-propagation through time needs
 * (MetaOptimizer.meta_minimize, DM/meta.py:398-414): for t = 0..T-1 the packed LSTM state
 * BEFORE step t, the gradient fed to the network at step t and (RNNProp) the moments AFTER
 * step t, plus the gradient at x_T.  hist == NULL is l2o_unroll. */
typedef struct l2o_unroll_hist {
  float* st;        /* device [T][l2o_state_floats(B_local, D)] */
  float* g;         /* device [T][B_local*D]                    */
  float* m;         /* device [T][B_local*D], RNNProp only      */
  float* v;
  float* g_final;   /* device [B_local*D]                       */
} l2o_unroll_hist;
int l2o_unroll_record(const l2o_net_cfg* cfg, const float* wpack /* device */,
                      const l2o_problem* prob, float* x, float* st, float* m, float* v,
                      int32_t T, int32_t step0, float* fx_part, void* workspace,
                      const l2o_unroll_hist* hist, void* stream);
/* The same unroll with the passes around it folded in (ABI v6); hist may be NULL:
 *   fx     also leaves fx[t] = (sum_b fx_part[t][b]) / B_global, t = 0..T (== l2o_reduce_fx, same summation order) --
 *          for the two-CU form inside the unroll's own epilogue kernel;
 *   x0     if not NULL the unroll starts from x0 [B_local, D] (read-only) and writes x_T to x: the `reset` of the
 *          iterate (DM/meta.py:379-383 re-runs the x initializer; a driver that restarts the SAME instance keeps x0);
 *   flags  L2O_UNROLL_ZERO_STATE: start from the zero LSTM state and zero RNNProp moments (what `reset` leaves,
 *          DM/meta.py:381, DM/meta_rnnprop_train.py:559-566) instead of reading st, m, v; they are still written.
 * i.e. `reset` + the first unroll of an epoch + fx_array.stack() in one call, without memset / copy passes. */
#define L2O_UNROLL_ZERO_STATE 1
int l2o_unroll_reduce(const l2o_net_cfg* cfg, const float* wpack /* device */, const l2o_problem* prob,
                      const float* x0 /* device or NULL */, float* x, float* st, float* m, float* v, int32_t T,
                      int32_t step0, int32_t flags, float* fx_part, float* fx /* device [T+1] */, void* workspace,
                      const l2o_unroll_hist* hist, void* stream);
/* Workspace contract (ABI v6).  The workspace of l2o_unroll must be ZERO before its first use and whenever
 * l2o_unroll_workspace_layout(cfg, prob) differs from the previous launch on it (another batch size / padded problem
 * size: the exchange granules move); l2o_unroll_workspace_init zeroes `bytes` bytes asynchronously (a hipMemsetAsync).
 * Between launches of one layout the library keeps the granule area clean itself (the epilogue kernel of a launch
 * re-zeroes it), so there is no memset per unroll.  layout == 0: the pair has no workspace-using kernel. */
int l2o_unroll_workspace_init(void* workspace, size_t bytes, void* stream);
int64_t l2o_unroll_workspace_layout(const l2o_net_cfg* cfg, const l2o_problem* prob);
int l2o_unroll_status(const void* workspace_header_host /* host copy of the first 4 bytes */);
/* 1 if l2o_unroll has a fused kernel for this (cfg, prob) pair, else 0: the LDS-resident forms
 * (D <= 128, M <= 16 ceil(D/16)) or the streaming form (everything else with D <= 512, D % 4 == 0,
 * any M: one workgroup per problem, the matrix streamed once per step, x / LSTM state / moments
 * on-chip). */
int l2o_unroll_supported(const l2o_net_cfg* cfg, const l2o_problem* prob);
/* 1 if l2o_unroll_record (hist != NULL) has a kernel for the pair: every fused form records (ABI v6: the
 * streaming form for D <= 512 too; v4 / v5: the LDS-resident forms only). */
int l2o_unroll_record_supported(const l2o_net_cfg* cfg, const l2o_problem* prob);

/* ---- fx_array.stack() / tf.reduce_mean over the batch (DM/meta.py:345, 374-376):
 * fx[t] = (sum_b fx_part[t*B_local + b]) / B_global, fixed summation order
 * (bit-reproducible).  With B sharded, all-reduce(sum) fx over ranks afterwards. */
int l2o_reduce_fx(const float* fx_part, int32_t T1, int32_t B_local, int32_t B_global,
                  float* fx /* device [T1] */, void* stream);

/* ---- small vector passes of the meta-gradient (ABI v11): what tf.gradients emits around the optimizer network when
 * MetaOptimizer.meta_minimize differentiates loss = sum_t f(x_t) (DM/meta.py:372-376, 398-414).  Until v10 the host
 * layer ran them as tensor arithmetic of its array library; the training path now calls l2o_* entry points only.
 *
 * l2o_suffix_sums: out[t][i] = g_final[i] + sum_{tau > t} g[tau][i], t = 0..T-1 -- dL/d(delta_t) with the optimizee
 *   gradients held constant (tf.stop_gradient, DM/meta.py:328-329).  g: DEVICE table of T device pointers to [n] floats
 *   (the recorded gradients live in per-step buffers), out: device [T][n].
 * l2o_colsum: out[b][k] (+)= sum_r A[b][r][k] for `batch` row-major [rows, cols] matrices -- bias gradients dz^T 1
 *   (DM/networks.py:192-203: b_gates, Linear b), the column sums of square_cos' wcos (DM/problems.py:959-994).  Fixed
 *   summation order.  scratch: device, >= l2o_colsum_scratch_floats(batch, cols) floats.
 * l2o_lincomb: out = ca a + cb b + cc c (b, c may be NULL; out may alias an input), n floats.
 * l2o_rnnprop_input_adjoint: second_derivatives=True for RNNProp (DM/meta_rnnprop_train.py:380-388 without the
 *   stop_gradient): from the BPTT step's du (H columns starting at column du_col of the Bm rows, leading dimension ldb)
 *   forms dg = dL/dg_t through (m~, g~) AND the moment recurrences, and updates the carried adjoints dm, dv in place.
 *   pow1 = beta1^k, pow2 = beta2^k, k = step + t; w_fc: device [2][H] (Sonnet layout of input_projection/w). */
int l2o_suffix_sums(const float* const* g /* device table */, const float* g_final, float* out, int64_t n, int32_t T,
                    void* stream);
size_t l2o_colsum_scratch_floats(int64_t batch, int32_t cols);
int l2o_colsum(const float* A, int64_t batch, int64_t rows, int32_t cols, float* out /* device [batch][cols] */,
               int32_t accumulate, float* scratch, void* stream);
int l2o_lincomb(float* out, const float* a, float ca, const float* b, float cb, const float* c, float cc, int64_t n,
                void* stream);
int l2o_rnnprop_input_adjoint(const float* Bm, int64_t ldb, int32_t du_col, int32_t H, const float* w_fc,
                              const float* g, const float* m, const float* v, double pow1, double pow2,
                              double beta1, double beta2, float* dm, float* dv, float* dg, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* L2O_ABI_H_ */
