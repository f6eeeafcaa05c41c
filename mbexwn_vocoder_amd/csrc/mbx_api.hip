// C ABI of the mel-inversion engine (include/mbexwn.h): handle, workspace carving, launch sequence.
//
// The launch sequence restates the inference branch of MBExWN.call
// (reference MBExWN_NVoc/vocoder/model/custom_pulsed_generator.py:556-771) driven the way
// PaNWaveNet.infer drives it (reference MBExWN_NVoc/vocoder/model/wavegen_1d.py:483-526).
// Nothing in here allocates, frees or synchronises after mbx_create: every call only enqueues kernels
// on the caller's stream, so a forward pass can be captured into a hipGraph by the caller.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mbexwn.h"
#include "mbx_kernels.h"

namespace {

thread_local std::string g_last_error;

mbx_status fail(mbx_status st, const std::string &msg) {
    g_last_error = msg;
    return st;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t err__ = (expr);                                                                     \
        if (err__ != hipSuccess)                                                                       \
            return fail(MBX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));            \
    } while (0)

struct DevTensor {
    float *ptr = nullptr;
    int ndim = 0;
    long long shape[4] = {0, 0, 0, 0};
    long long count = 0;
};

struct StageRef {
    const void *ptr = nullptr;
    long long count = 0, stride = 0;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Makes the handle's device current for the duration of a call and puts the caller's device back afterwards, so
// that an engine can be created for / used from a thread whose current device is another one.
struct DeviceGuard {
    int prev = -1;
    bool switched = false, ok = true;
    explicit DeviceGuard(int device) {
        ok = hipGetDevice(&prev) == hipSuccess;
        if (ok && prev != device) {
            ok = hipSetDevice(device) == hipSuccess;
            switched = ok;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace

// stages of the launch sequence that mbx_profile_read can report (names in kProfNames)
enum { PROF_GATE = 0, PROF_RES_SKIP, PROF_FRONTEND, PROF_WAVETABLE, PROF_START, PROF_TAIL, PROF_PQMF, PROF_STFT_FILTER,
       PROF_OVERLAP_ADD, PROF_NORM_MEL, PROF_GATE0, PROF_RES_SKIP_F16, PROF_KINDS };
static const char *const kProfNames[PROF_KINDS] = {"gate", "res_skip", "frontend", "wavetable", "start", "tail", "pqmf",
                                                   "stft_filter", "overlap_add", "norm_mel", "gate0", "res_skip_f16"};

struct mbx_handle {
    mbx_config cfg;
    int device = 0;
    char *arena = nullptr;
    size_t arena_bytes = 0;
    std::map<std::string, DevTensor> tensors;
    std::map<int, std::pair<float *, float *>> lerp;   // interpolation factor -> (w0, w1)
    float *twiddle = nullptr;
    float *zeros = nullptr;   // 256 bytes of zeros (padding source of the LDS-DMA GEMMs)
    float *poly = nullptr;
    float *poly_t = nullptr;          // the same table as the MFMA B operand: (4 * ceil(K / 4), 16), K = poly_ndm * subbands, zero padded
    int poly_ndm = 0, poly_dm_min = 0;
    std::map<std::string, StageRef> stages;
    // derived
    int f0_time_factor = 1, vtf_time_factor = 1;
    long long subnet_buf_per_frame = 0;   // floats per frame of one ping-pong buffer
    int last_gate_kernel[MBX_MAX_WN_LAYERS] = {};   // MBX_GATE_K_* of the most recent forward (mbx_conv_form_info.gate_kernel)
    int last_gate_layers = 0;
    bool f0_full64 = false;               // mbx_config.f0_accumulate == MBX_F0_ACC_F64 and the F0-net has the shape (conv [prelu | leaky])* head
                                          // with its "<layer>.w64" tensors: float64 weights and hidden layers (f0_chain_is_full64)
    std::vector<mbx_subnet_op> cond_ops;  // pre-conditioning convolutions + the conditioning layer (empty: conditioning disabled)
    long long cond_buf_per_frame = 0;     // floats per frame of a ping-pong buffer of that chain (0: no pre-conditioning layers)
    // several WaveNet blocks (mbx_config.n_wn_blocks > 1; empty: the single-block path)
    struct WnBlock {
        int C = 0, ups = 1, spf = 0, ccu = 0;      // channels, upsampling factor behind the block, rows per frame, conditioning rows per frame
        std::string prefix;                        // "wn." | "wn1." ...
        std::vector<mbx_subnet_op> cond_ops;       // its pre-conditioning + conditioning chain (empty: conditioning disabled)
    };
    std::vector<WnBlock> blocks;
    long long mb_hc_per_frame = 0;                 // max over the blocks of rows per frame x channels
    bool fold_skip = false;      // skip path folded into the end convolution (needs the *.fold tensors)
    bool fold_start = false;     // start convolution folded into layer 0 (needs fold_skip and the *.start_fold / *.fold_start tensors)
    bool winograd4_always = false;   // mbx_config.batch_invariant with F(4,3): the large-launch kernel shapes at every size
    int gate_small_shape = -1;       // mbx_config.tune_gate_shape: pins the F(4,3) block shape of small launches (0: 256-row | 1: product-split | 2: product-split, half column tiles; same bits)
    long long resskip_wave_tiles = 2048;   // default policy: res/skip launches of at most this many 16-row tiles run the wave-tiled kernel
    int resskip_split = 0;           // mbx_config.tune_resskip_split
    bool split_f16 = false;          // mbx_config.wn_precision == MBX_PRECISION_SPLIT_F16 and the images are there
    bool split_f16_gate = false;     // ... for the gate layers too (wn_gate_f16.hip)
    float calib_err_split = -1.f;    // max |audio(split precision) - audio(float32 direct form)| of the calibration run
    int split_rejected = 0;          // the calibration switched the split precision off (error above the threshold, or not finite)
    int winograd = 0;            // gate layer form in effect: 0 direct, 2 Winograd F(2,3), 4 Winograd F(4,3) (needs the packed weights)
    // what mbx_conv_form reports
    int calibrated = 0;
    float calib_err43 = -1.f, calib_err23 = -1.f, calib_ref = 0.f, calib_threshold = 0.f;
    // bench-only kernel timing (mbx_profile_*): one event pool per stage of the launch sequence
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool[PROF_KINDS];
    size_t ev_used[PROF_KINDS] = {};
};

namespace {

const DevTensor *find(const mbx_handle *h, const std::string &name) {
    auto it = h->tensors.find(name);
    return it == h->tensors.end() ? nullptr : &it->second;
}

// floats per mel frame needed by the widest intermediate of a sub-net, and its time factor
mbx_status analyse_subnet(const mbx_subnet_op *ops, int n_ops, int cin, long long *per_frame, int *factor,
                          int *cout) {
    long long fac = 1, chan = cin, widest = cin;
    for (int i = 0; i < n_ops; ++i) {
        const mbx_subnet_op &op = ops[i];
        if (op.kind == MBX_OP_CONV) {
            if (op.cin != chan) return fail(MBX_ERR_INVALID_ARGUMENT, std::string("sub-net op ") + op.name + ": cin mismatch");
            if (op.up < 1 || op.cout % op.up) return fail(MBX_ERR_INVALID_ARGUMENT, "sub-pixel factor must divide cout");
            widest = std::max(widest, fac * op.cout);
            chan = op.cout / op.up;
            fac *= op.up;
        } else if (op.kind == MBX_OP_LIN) {
            fac *= op.up;
            widest = std::max(widest, fac * chan);
        }
    }
    *per_frame = widest;
    *factor = (int)fac;
    *cout = (int)chan;
    return MBX_OK;
}

struct Workspace {
    float *mel_norm, *nm_a, *nm_b;
    double *f0h0, *f0h1;   // float64 hidden layers of the F0-net (mbx_handle::f0_full64)
    float *sub0, *sub1, *sub2, *sub3, *sub4, *sub5, *f0_wide, *f0, *cum, *chunk_last, *pulse, *cond, *h, *a, *skip, *wn_out, *sub, *exc, *ceps, *frames;
    float *mb_h, *mb_a, *mb_skip, *mb_y0, *mb_y1, *mb_cond[MBX_MAX_WN_BLOCKS];   // several WaveNet blocks only
    float *pulse_ana;   // PQMF analysis of the pulse signal (pulse_pqmf_taps > 0): the WaveNet's excitation rows
    float *h16;         // split half precision: the hidden state as fp16 planes (ConvArgs::h_split), round_up(C, 8) words per row
    int *ceps_index;
    size_t total;
};

Workspace carve(const mbx_handle *hd, char *base, int B, int T) {
    const mbx_config &c = hd->cfg;
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t n_floats) {
        char *p = base ? base + off : nullptr;
        off += align_up(n_floats * sizeof(float), 256);
        return reinterpret_cast<float *>(p);
    };
    const size_t BT = (size_t)B * T;
    const size_t npulse = (size_t)T * c.pulse_per_frame, nsteps = (size_t)T * c.steps_per_frame;
    const int chunks = (int)((npulse + c.phase_chunk - 1) / c.phase_chunk) + 1;
    const bool nm = c.nm_iters > 0;
    w.mel_norm = take((nm || hd->f0_full64) ? BT * c.mel_channels : 0);    // (also: aligned copy of a misaligned mel for the float64 F0 chain)
    w.nm_a = take(nm ? BT : 0);
    w.nm_b = take(nm ? BT : 0);
    w.sub0 = take(BT * hd->subnet_buf_per_frame);
    w.sub1 = take(BT * hd->subnet_buf_per_frame);
    w.sub2 = take(BT * hd->subnet_buf_per_frame);   // VTF-net ping-pong (its convolutions share launches with the F0-net's)
    w.sub3 = take(BT * hd->subnet_buf_per_frame);
    w.f0h0 = reinterpret_cast<double *>(take(hd->f0_full64 ? 2 * BT * hd->subnet_buf_per_frame : 0));
    w.f0h1 = reinterpret_cast<double *>(take(hd->f0_full64 ? 2 * BT * hd->subnet_buf_per_frame : 0));
    w.sub4 = take(BT * hd->cond_buf_per_frame);     // pre-conditioning layers (usually none: zero floats)
    w.sub5 = take(BT * hd->cond_buf_per_frame);
    // an F0-net with a bare ["L", up] entry runs at a multiple of the pulse rate and is cut to it (reference
    // custom_pulsed_generator.py:57-60, 787): the uncut contour lives here
    w.f0_wide = take(hd->f0_time_factor > c.pulse_per_frame ? BT * hd->f0_time_factor : 0);
    w.f0 = take(B * npulse);
    w.cum = take(B * npulse);
    w.chunk_last = take((size_t)B * chunks);
    w.pulse = take(B * npulse * (1 + c.wt_subharm_channels));
    w.pulse_ana = take(c.pulse_pqmf_taps > 0 ? B * npulse : 0);
    w.cond = take(BT * 2 * c.wn_channels * c.cond_conv_upsampling);
    w.h = take(B * nsteps * c.wn_channels);
    w.a = take(B * nsteps * (c.wn_channels + 16));   // layer 0 appends the excitation channels to its rows (wn_gate0.hip)
    w.skip = take(B * nsteps * c.wn_channels);
    w.h16 = take(hd->split_f16_gate ? B * nsteps * (size_t)((c.wn_channels + 7) / 8 * 8) : 0);
    w.wn_out = take(B * nsteps * c.wn_out_channels);
    w.sub = take(B * nsteps * c.subbands);
    w.exc = take(BT * c.hop_size);
    w.ceps = take(BT * c.n_ceps);
    w.ceps_index = reinterpret_cast<int *>(take(BT));
    w.frames = take(BT * c.stft_win);
    const bool mb = !hd->blocks.empty();
    w.mb_h = take(mb ? BT * hd->mb_hc_per_frame : 0);
    w.mb_a = take(mb ? BT * hd->mb_hc_per_frame : 0);
    w.mb_skip = take(mb ? BT * hd->mb_hc_per_frame : 0);
    w.mb_y0 = take(mb ? B * nsteps * c.wn_out_channels : 0);
    w.mb_y1 = take(mb ? B * nsteps * c.wn_out_channels : 0);
    for (int b = 0; b < MBX_MAX_WN_BLOCKS; ++b)
        w.mb_cond[b] = take(mb && b >= 1 && b < (int)hd->blocks.size() ? BT * hd->blocks[b].ccu * 2 * hd->blocks[b].C : 0);
    w.total = off;
    return w;
}

mbx::ConvArgs conv_args(const float *x, long long x_bstride, int ldx, const int *n_frames, int rpf, int max_rows,
                        int batch, const DevTensor *w, const DevTensor *bias, int ks, int cin, int cout, int dil,
                        int pad_l, int pad_mode, float *out, long long out_bstride, int ldo) {
    mbx::ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x;
    a.x_bstride = x_bstride;
    a.ldx = ldx;
    a.n_frames = n_frames;
    a.rows_per_frame = rpf;
    a.max_rows = max_rows;
    a.batch = batch;
    a.w = w->ptr;
    a.bias = bias ? bias->ptr : nullptr;
    a.cin = cin;
    a.cout = cout;
    a.ks = ks;
    a.dil = dil;
    a.pad_l = pad_l;
    a.pad_mode = pad_mode;
    a.out = out;
    a.out_bstride = out_bstride;
    a.ldo = ldo;
    return a;
}

// Head of the F0-net: [conv 1x1 -> 1 channel] [lin] ([act]) at the end of the op list (reference
// custom_pulsed_generator.py:126-146): one float64 kernel (launch_f0_head) under mbx_config.f0_accumulate == MBX_F0_ACC_F64
bool is_f0_head(const mbx_subnet_op *ops, int n_ops, int i) {
    const int n_tail = n_ops - i;
    return ops[i].kind == MBX_OP_CONV && ops[i].ks == 1 && ops[i].cout == 1 && ops[i].up == 1 && (n_tail == 2 || n_tail == 3) &&
           ops[i + 1].kind == MBX_OP_LIN && (n_tail == 2 || ops[i + 2].kind == MBX_OP_ACT);
}

const double *f64_weights(const mbx_handle *hd, const mbx_subnet_op &op) {
    const DevTensor *t = find(hd, std::string(op.name) + ".w64");
    if (!t || t->count != 2LL * op.ks * op.cin * op.cout || (reinterpret_cast<uintptr_t>(t->ptr) & 7)) return nullptr;
    return reinterpret_cast<const double *>(t->ptr);
}

// The whole F0-net in float64 (weights, hidden layers, head): its op list is (conv [prelu | leaky])* head, every layer has
// its float64 weights "<layer>.w64" and rows of whole float4 / double4 groups.  Other shapes of the grammar keep float32
// weights and hidden layers and accumulate in float64 (ConvArgs::precise alone).
bool f0_chain_is_full64(const mbx_handle *hd) {
    const mbx_config &c = hd->cfg;
    if (c.f0_accumulate != MBX_F0_ACC_F64 || c.n_f0_ops < 3) return false;
    int k = 0;
    while (k < c.n_f0_ops) {
        const mbx_subnet_op &op = c.f0_ops[k];
        if (op.kind != MBX_OP_CONV || op.up != 1 || op.cin % 4 || !f64_weights(hd, op)) return false;
        if (is_f0_head(c.f0_ops, c.n_f0_ops, k)) return k > 0;
        if (op.cout % 4) return false;
        ++k;
        if (k < c.n_f0_ops && (c.f0_ops[k].kind == MBX_OP_PRELU || c.f0_ops[k].kind == MBX_OP_LEAKY)) ++k;
    }
    return false;
}

// Executes a sub-net op list (reference custom_pulsed_generator.py:38-148 flattened by the host).
// in (B, T, cin) -> final (B, T*factor, cout); optional affine y*scale+offset applied after the last op.
// Resumable: next_conv() launches the element-wise ops up to the next convolution and hands that convolution back
// un-launched, so that the caller can put the convolutions of independent sub-nets into one launch
// (launch_conv1d_group); the caller launches it before calling next_conv() again.
struct SubnetRun {
    mbx_handle *hd;
    const mbx_subnet_op *ops;
    int n_ops;
    const int *n_frames;
    int B, T;
    float *buf0, *buf1, *final_out;
    float scale, offset;
    hipStream_t stream;
    const float *cur;
    int chan, rpf = 1, pp = 0, last_writer = -1, i = 0;
    long long cur_bstride;
    long long stride_frames = 0;   // > 0: frames between batch items of the input and of the final output (a sub-window of a
                                   // longer window is being computed: mbx_forward_options.fe_new_frames); 0: T
    bool affine_done, finished = false;
    bool precise = false;          // the F0-net under mbx_config.f0_accumulate == MBX_F0_ACC_F64: float64 accumulation in its
                                   // convolutions (ConvArgs::precise) and its head -- final 1x1 convolution to one channel,
                                   // interpolation, final activation, affine map -- as one float64 kernel
    double *buf64_0 = nullptr, *buf64_1 = nullptr;   // ... with float64 weights and hidden layers (mbx_handle::f0_full64)
    const double *cur64 = nullptr;                   // the current tensor when it is a float64 one (cur is null then)
    mbx_status status = MBX_OK;
    void window_stride(long long frames, int cin) {
        stride_frames = frames;
        cur_bstride = frames * cin;
    }

    SubnetRun(mbx_handle *hd_, const mbx_subnet_op *ops_, int n_ops_, const float *in, int cin, const int *n_frames_,
              int B_, int T_, float *buf0_, float *buf1_, float *final_out_, bool affine, float scale_, float offset_,
              hipStream_t stream_)
        : hd(hd_), ops(ops_), n_ops(n_ops_), n_frames(n_frames_), B(B_), T(T_), buf0(buf0_), buf1(buf1_),
          final_out(final_out_), scale(scale_), offset(offset_), stream(stream_), cur(in), chan(cin),
          cur_bstride((long long)T_ * cin), affine_done(!affine) {
        // index of the last op that launches a kernel writing a new buffer
        for (int k = 0; k < n_ops; ++k)
            if (ops[k].kind == MBX_OP_CONV || ops[k].kind == MBX_OP_LIN) last_writer = k;
    }
    bool stop(mbx_status st) {
        status = st;
        finished = true;
        return false;
    }
    bool next_conv(mbx::ConvArgs &pending) {
        if (finished) return false;
        for (; i < n_ops; ++i) {
            const mbx_subnet_op &op = ops[i];
            if (op.kind == MBX_OP_CONV) {
                const DevTensor *w = find(hd, std::string(op.name) + ".w");
                const DevTensor *bias = find(hd, std::string(op.name) + ".b");
                if (!w || !bias) return stop(fail(MBX_ERR_INVALID_ARGUMENT, std::string("missing tensor ") + op.name + ".w/.b"));
                // float64 head: [conv 1x1 -> 1 channel] [lin] ([act]) at the end of the list
                const int n_tail = n_ops - i;
                if (precise && is_f0_head(ops, n_ops, i)) {
                    const mbx_subnet_op &lin = ops[i + 1];
                    auto it = hd->lerp.find(lin.up);
                    if (it == hd->lerp.end()) return stop(fail(MBX_ERR_INVALID_ARGUMENT, "interpolation table missing"));
                    const long long out_bstride = (stride_frames ? stride_frames : (long long)T) * rpf * lin.up;
                    mbx::launch_f0_head(cur, cur64, cur_bstride, op.cin, n_frames, rpf, T * rpf, B, w->ptr,
                                        buf64_0 ? f64_weights(hd, op) : nullptr, bias->ptr, lin.up,
                                        it->second.first, it->second.second, n_tail == 3 ? ops[i + 2].act : MBX_ACT_LINEAR,
                                        affine_done ? 1.f : scale, affine_done ? 0.f : offset, final_out, out_bstride, stream);
                    cur64 = nullptr;
                    affine_done = true;
                    cur = final_out;
                    chan = 1;
                    rpf *= lin.up;
                    cur_bstride = out_bstride;
                    i = n_ops;
                    break;
                }
                float *out = (i == last_writer) ? final_out : (pp ? buf1 : buf0);
                pp ^= 1;
                const long long out_bstride = ((i == last_writer && stride_frames) ? stride_frames : (long long)T) * rpf * op.cout;
                mbx::ConvArgs a = conv_args(cur, cur_bstride, chan, n_frames, rpf, T * rpf, B, w, bias, op.ks, op.cin,
                                            op.cout, 1, op.pad_l, op.pad_mode, out, out_bstride, op.cout);
                a.precise = precise ? 1 : 0;
                a.zeros = hd->zeros;
                if (buf64_0) {             // full-float64 chain (the op list was checked at mbx_create: f0_chain_is_full64)
                    double *out64 = (pp ^ 1) ? buf64_1 : buf64_0;       // (pp was toggled above)
                    a.w64 = f64_weights(hd, op);
                    a.x64 = cur64;
                    a.out64 = out64;
                    if (cur64) a.x = nullptr;
                    a.out = nullptr;
                    cur64 = out64;
                    out = nullptr;
                }
                if (i + 1 < n_ops && ops[i + 1].kind == MBX_OP_PRELU && op.up == 1) {
                    const DevTensor *al = find(hd, std::string(ops[i + 1].name) + ".alpha");
                    if (!al) return stop(fail(MBX_ERR_INVALID_ARGUMENT, std::string("missing tensor ") + ops[i + 1].name + ".alpha"));
                    a.alpha = al->ptr;
                    ++i;
                } else if (i + 1 < n_ops && ops[i + 1].kind == MBX_OP_LEAKY) {
                    a.use_leaky = 1;
                    a.leaky = ops[i + 1].alpha;
                    ++i;
                }
                cur = out;
                chan = op.cout / op.up;
                rpf *= op.up;
                cur_bstride = out_bstride;
                ++i;
                pending = a;
                return true;
            } else if (op.kind == MBX_OP_LIN) {
                auto it = hd->lerp.find(op.up);
                if (it == hd->lerp.end()) return stop(fail(MBX_ERR_INVALID_ARGUMENT, "interpolation table missing"));
                float *out = (i == last_writer) ? final_out : (pp ? buf1 : buf0);
                pp ^= 1;
                int act = MBX_ACT_LINEAR;
                float sc = 1.f, of = 0.f;
                int consumed = 0;
                if (i + 1 < n_ops && ops[i + 1].kind == MBX_OP_ACT) {
                    act = ops[i + 1].act;
                    consumed = 1;
                }
                if (i + consumed == n_ops - 1 && !affine_done) {
                    sc = scale;
                    of = offset;
                    affine_done = true;
                }
                const long long out_bstride = ((i == last_writer && stride_frames) ? stride_frames : (long long)T) * rpf * op.up * chan;
                mbx::launch_lin_interp(cur, cur_bstride, n_frames, rpf, T * rpf, B, chan, op.up, it->second.first,
                                       it->second.second, act, sc, of, out, out_bstride, stream);
                i += consumed;
                cur = out;
                rpf *= op.up;
                cur_bstride = out_bstride;
            } else if (op.kind == MBX_OP_PRELU || op.kind == MBX_OP_LEAKY) {
                const DevTensor *al = op.kind == MBX_OP_PRELU ? find(hd, std::string(op.name) + ".alpha") : nullptr;
                if (op.kind == MBX_OP_PRELU && !al) return stop(fail(MBX_ERR_INVALID_ARGUMENT, "missing PReLU slopes"));
                mbx::launch_prelu(const_cast<float *>(cur), cur_bstride, n_frames, rpf, T * rpf, B, chan,
                                  al ? al->ptr : nullptr, op.alpha, stream);
            } else if (op.kind == MBX_OP_ACT) {
                float sc = 1.f, of = 0.f;
                if (i == n_ops - 1 && !affine_done) {
                    sc = scale;
                    of = offset;
                    affine_done = true;
                }
                mbx::launch_activation(cur, cur_bstride, n_frames, rpf, T * rpf, B, chan, op.act, sc, of,
                                       const_cast<float *>(cur), cur_bstride, stream);
            } else {
                return stop(fail(MBX_ERR_INVALID_ARGUMENT, "unknown sub-net op kind"));
            }
        }
        if (!affine_done)
            mbx::launch_activation(cur, cur_bstride, n_frames, rpf, T * rpf, B, chan, MBX_ACT_LINEAR, scale, offset,
                                   const_cast<float *>(cur), cur_bstride, stream);
        if (cur != final_out) return stop(fail(MBX_ERR_INVALID_ARGUMENT, "sub-net without a convolution"));
        finished = true;
        return false;
    }
};

// brackets one launch with events when profiling is on
struct ScopedEvents {
    mbx_handle *hd;
    int kind;
    hipStream_t stream;
    hipEvent_t stop = nullptr;
    ScopedEvents(mbx_handle *h, int k, hipStream_t s) : hd(h), kind(k), stream(s) {
        if (!hd->profiling) return;
        auto &pool = hd->ev_pool[kind];
        if (hd->ev_used[kind] == pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            pool.push_back({a, b});
        }
        auto &pr = pool[hd->ev_used[kind]++];
        (void)hipEventRecord(pr.first, stream);
        stop = pr.second;
    }
    ~ScopedEvents() {
        if (stop) (void)hipEventRecord(stop, stream);
    }
};

mbx::WaveTableConsts wavetable_consts(const mbx_handle *hd) {
    const mbx_config &c = hd->cfg;
    mbx::WaveTableConsts k;
    k.tables = find(hd, "table.wavetables")->ptr;
    k.n_period = c.wt_n_period;
    k.n_tables = c.wt_n_tables;
    k.pulse_rate = c.pulse_rate;
    k.nominal_f0 = c.wt_nominal_f0;
    k.min_tf = c.wt_min_transposition;
    k.max_tf = c.wt_max_transposition;
    k.grid_norm = c.wt_grid_norm;
    k.chunk = c.phase_chunk;
    k.n_sub = c.wt_subharm_channels;
    k.sin_fun = c.wt_sinusoid_as_fun;
    return k;
}

mbx::NormMelConsts norm_mel_consts(const mbx_handle *hd) {
    const mbx_config &c = hd->cfg;
    mbx::NormMelConsts k;
    k.iters = c.nm_iters;
    k.mel_channels = c.mel_channels;
    k.hop = c.hop_size;
    k.win = c.stft_win;
    k.smooth_win = c.nm_smooth_win;
    k.cut = c.nm_smooth_win / 2 + 2 * c.hop_size - c.stft_win / 2;
    k.rms_norm_fact = c.nm_rms_norm_fact;
    k.rms_floor = c.nm_rms_floor;
    k.compressor_exp = c.nm_compressor_exp;
    k.lin_amp_scale = c.nm_lin_amp_scale;
    k.lin_amp_off = c.nm_lin_amp_off;
    k.mel_amp_scale = c.nm_mel_amp_scale;
    k.use_compressor = c.nm_use_compressor;
    k.use_max_limit = c.nm_use_max_limit;
    k.inv_enorm = find(hd, "table.nm_inv_enorm")->ptr;
    k.pinv = c.nm_use_pinv ? find(hd, "table.nm_pinv")->ptr : nullptr;
    k.n_bins = c.fft_size / 2 + 1;
    k.win_norm = c.nm_win_norm;
    k.gwin = find(hd, "table.nm_gwin")->ptr;
    k.smooth_win_table = find(hd, "table.nm_smooth_win")->ptr;
    return k;
}

mbx::StftConsts stft_consts(const mbx_handle *hd) {
    const mbx_config &c = hd->cfg;
    mbx::StftConsts k;
    k.hop = c.hop_size;
    k.win = c.stft_win;
    k.fft_size = c.fft_size;
    k.n_ceps = c.n_ceps;
    k.n_ceps_windows = c.n_ceps_windows;
    k.max_log_range = c.filter_max_log_range;
    k.preserve_energy = c.spect_preserve_energy;
    k.hann = find(hd, "table.hann")->ptr;
    k.inv_win = find(hd, "table.inv_win")->ptr;
    k.twiddle = hd->twiddle;
    k.ceps_windows = c.n_ceps_windows ? find(hd, "table.ceps_windows")->ptr : nullptr;
    k.ceps_log10f0 = c.n_ceps_windows ? find(hd, "table.ceps_log10f0")->ptr : nullptr;
    k.f0_smooth = c.n_ceps_windows ? find(hd, "table.f0_smooth")->ptr : nullptr;
    k.pulse_per_frame = c.pulse_per_frame;
    return k;
}

}  // namespace

extern "C" {

// the form of the dilated convolution (mbx_config.wn_conv_form) and its calibration: defined behind forward_impl
static bool form_available(const mbx_handle *hd, int form);
static void set_form(mbx_handle *hd, int form);
static mbx_status calibrate_on_synthetic_mel(mbx_handle *hd, bool forms);

const char *mbx_last_error(void) { return g_last_error.c_str(); }

mbx_status mbx_create(const mbx_config *config, const mbx_tensor *tensors, int32_t n_tensors, int32_t device,
                      mbx_handle **out) {
    if (!config || !tensors || !out) return fail(MBX_ERR_INVALID_ARGUMENT, "null argument");
    if (config->struct_size != (int32_t)sizeof(mbx_config) || config->abi_version != MBX_ABI_VERSION)
        return fail(MBX_ERR_INVALID_ARGUMENT, "mbx_config ABI mismatch (struct_size / abi_version)");
    const mbx_config &c = *config;
    if (c.wn_layers < 1 || c.wn_layers > MBX_MAX_WN_LAYERS) return fail(MBX_ERR_INVALID_ARGUMENT, "wn_layers out of range");
    if (c.n_f0_ops < 1 || c.n_f0_ops > MBX_MAX_SUBNET_OPS || c.n_vtf_ops < (c.ps_off ? 0 : 1) || c.n_vtf_ops > MBX_MAX_SUBNET_OPS ||
        (c.ps_off && c.n_vtf_ops != 0))
        return fail(MBX_ERR_INVALID_ARGUMENT, "sub-net op count out of range (ps_off: no VTF-net)");
    if (c.wn_channels % 4 || c.wn_kernel_size % 2 != 1) return fail(MBX_ERR_INVALID_ARGUMENT, "wn_channels must be a multiple of 4, kernel size odd");
    if (c.fft_size > 2048 || (c.fft_size & (c.fft_size - 1)) || c.stft_win > c.fft_size || c.stft_win != 4 * c.hop_size)
        return fail(MBX_ERR_UNSUPPORTED, "STFT geometry: need power-of-two fft_size <= 2048 and win == 4*hop");
    if (c.hop_size % c.subbands || c.steps_per_frame * c.subbands != c.hop_size)
        return fail(MBX_ERR_INVALID_ARGUMENT, "hop_size must be steps_per_frame * subbands");
    // rows per frame of the first WaveNet block: the sub-band rate divided by the in-block upsampling factors
    int spf0 = c.steps_per_frame;
    if (c.n_wn_blocks > MBX_MAX_WN_BLOCKS || c.n_wn_blocks < 0) return fail(MBX_ERR_INVALID_ARGUMENT, "n_wn_blocks out of range");
    if (c.n_wn_blocks >= 1) {
        if (c.wn_block_channels[0] != c.wn_channels) return fail(MBX_ERR_INVALID_ARGUMENT, "wn_block_channels[0] must be wn_channels");
        for (int b = 0; b < c.n_wn_blocks; ++b) {
            if (c.wn_block_ups[b] < 1 || c.wn_block_channels[b] < 4 || c.wn_block_channels[b] % 4 || spf0 % c.wn_block_ups[b])
                return fail(MBX_ERR_INVALID_ARGUMENT, "WaveNet blocks: channels must be multiples of 4, upsampling factors must divide steps_per_frame");
            spf0 /= c.wn_block_ups[b];
        }
    }
    if (spf0 * c.pulse_channels != c.pulse_per_frame)
        return fail(MBX_ERR_INVALID_ARGUMENT, "pulse_per_frame must be (rows per frame of the first WaveNet block) * pulse_channels");
    if ((spf0 % c.cond_lin_upsampling) || spf0 / c.cond_lin_upsampling != c.cond_conv_upsampling)
        return fail(MBX_ERR_INVALID_ARGUMENT, "conditioning rates do not reach the WaveNet rate");
    if (c.wt_subharm_channels < 0 || c.wt_subharm_channels > 8) return fail(MBX_ERR_INVALID_ARGUMENT, "wt_subharm_channels out of range");
    if (c.wn_in_channels != c.pulse_channels * (1 + c.wt_subharm_channels) + (c.noise_sigma != 0.f ? 1 : 0))
        return fail(MBX_ERR_INVALID_ARGUMENT, "wn_in_channels must be pulse_channels * (1 + wt_subharm_channels) (+1 with noise)");
    if (c.ps_subband_gain && (c.n_ceps != c.subbands || c.ps_off || c.n_ceps_windows))
        return fail(MBX_ERR_INVALID_ARGUMENT, "ps_subband_gain: the VTF-net ends in one gain per sub-band (n_ceps == subbands), no lifter, not ps_off");
    if (c.pqmf_taps % 2) return fail(MBX_ERR_INVALID_ARGUMENT, "PQMF taps must be even");
    if (c.pulse_pqmf_taps < 0 || c.pulse_pqmf_taps % 2 || (c.pulse_pqmf_taps > 0 && c.wt_subharm_channels))
        return fail(MBX_ERR_INVALID_ARGUMENT, "pulse_pqmf_taps must be even and >= 0, and excludes wt_subharm_channels");
    if (c.phase_chunk < 1 || c.phase_chunk > 1024) return fail(MBX_ERR_INVALID_ARGUMENT, "phase_chunk must be in [1, 1024]");
    if (c.wn_gate_activation < MBX_GATE_GTU || c.wn_gate_activation > MBX_GATE_GLU)
        return fail(MBX_ERR_INVALID_ARGUMENT, "wn_gate_activation must be MBX_GATE_GTU, MBX_GATE_GFU, MBX_GATE_GSU or MBX_GATE_GLU");
    if (c.n_precond < 0 || c.n_precond > MBX_MAX_PRECOND) return fail(MBX_ERR_INVALID_ARGUMENT, "n_precond out of range");
    for (int i = 0; i < c.n_precond; ++i)
        if (c.precond_channels[i] < 1) return fail(MBX_ERR_INVALID_ARGUMENT, "precond_channels must be positive");

    mbx_handle *hd = new mbx_handle();
    hd->cfg = c;
    hd->device = device;
    auto bail = [&](mbx_status st) {
        mbx_destroy(hd);
        return st;
    };
    DeviceGuard guard(device);
    if (!guard.ok) {
        delete hd;
        return fail(MBX_ERR_HIP, "hipSetDevice: cannot select device " + std::to_string(device));
    }
    hipError_t e = hipSuccess;

    // interpolation factors in use
    std::vector<int> ups = {c.cond_lin_upsampling};
    if (c.ps_subband_gain) ups.push_back(c.hop_size);     // the sub-band gains are interpolated by hop_size
    for (int i = 0; i < c.n_f0_ops; ++i)
        if (c.f0_ops[i].kind == MBX_OP_LIN) ups.push_back(c.f0_ops[i].up);
    for (int i = 0; i < c.n_vtf_ops; ++i)
        if (c.vtf_ops[i].kind == MBX_OP_LIN) ups.push_back(c.vtf_ops[i].up);

    // polyphase table of the PQMF synthesis bank
    const mbx_tensor *syn = nullptr;
    size_t total = 0;
    for (int i = 0; i < n_tensors; ++i) {
        long long cnt = 1;
        if (tensors[i].ndim < 1 || tensors[i].ndim > 4 || !tensors[i].data || !tensors[i].name)
            return bail(fail(MBX_ERR_INVALID_ARGUMENT, "malformed tensor entry"));
        for (int d = 0; d < tensors[i].ndim; ++d) cnt *= tensors[i].shape[d];
        total += align_up((size_t)cnt * sizeof(float), 256);
        if (std::strcmp(tensors[i].name, "table.pqmf_syn") == 0) syn = &tensors[i];
    }
    if (!syn || syn->ndim != 2 || syn->shape[0] != c.pqmf_taps + 1 || syn->shape[1] != c.subbands)
        return bail(fail(MBX_ERR_INVALID_ARGUMENT, "table.pqmf_syn must be (taps+1, subbands)"));
    const int M = c.subbands, half = c.pqmf_taps / 2;
    hd->poly_dm_min = -((half + M - 1) / M);
    const int dm_max = (half + M - 1) / M;
    hd->poly_ndm = dm_max - hd->poly_dm_min + 1;
    std::vector<float> poly((size_t)M * hd->poly_ndm * M, 0.f);
    for (int p = 0; p < M; ++p)
        for (int i = 0; i < hd->poly_ndm; ++i) {
            const int j = (hd->poly_dm_min + i) * M + half - p;
            if (j >= 0 && j <= c.pqmf_taps)
                for (int k = 0; k < M; ++k) poly[((size_t)p * hd->poly_ndm + i) * M + k] = syn->data[(size_t)j * M + k];
        }
    const int poly_k = hd->poly_ndm * M, poly_kpad = (poly_k + 3) / 4 * 4;
    std::vector<float> poly_t(M <= 16 ? (size_t)poly_kpad * 16 : 0, 0.f);
    if (M <= 16)
        for (int p = 0; p < M; ++p)
            for (int i = 0; i < poly_k; ++i) poly_t[(size_t)i * 16 + p] = poly[(size_t)p * poly_k + i];
    std::vector<float> tw((size_t)c.fft_size);
    for (int k = 0; k < c.fft_size / 2; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)c.fft_size;
        tw[2 * k] = (float)std::cos(ang);
        tw[2 * k + 1] = (float)std::sin(ang);
    }
    total += align_up(poly.size() * sizeof(float), 256) + align_up(poly_t.size() * sizeof(float), 256) +
             align_up(tw.size() * sizeof(float), 256) + 256;
    for (int u : ups) total += 2 * align_up((size_t)u * sizeof(float), 256);

    e = hipMalloc(reinterpret_cast<void **>(&hd->arena), total);
    if (e != hipSuccess) return bail(fail(MBX_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)));
    hd->arena_bytes = total;
    size_t off = 0;
    auto upload = [&](const float *src, size_t count) -> float * {
        float *dst = reinterpret_cast<float *>(hd->arena + off);
        off += align_up(count * sizeof(float), 256);
        hipError_t ee = hipMemcpy(dst, src, count * sizeof(float), hipMemcpyHostToDevice);
        return ee == hipSuccess ? dst : nullptr;
    };
    for (int i = 0; i < n_tensors; ++i) {
        DevTensor t;
        t.ndim = tensors[i].ndim;
        t.count = 1;
        for (int d = 0; d < t.ndim; ++d) {
            t.shape[d] = tensors[i].shape[d];
            t.count *= t.shape[d];
        }
        t.ptr = upload(tensors[i].data, (size_t)t.count);
        if (!t.ptr) return bail(fail(MBX_ERR_HIP, "hipMemcpy of a tensor failed"));
        hd->tensors[tensors[i].name] = t;
    }
    hd->poly = upload(poly.data(), poly.size());
    if (!poly_t.empty()) hd->poly_t = upload(poly_t.data(), poly_t.size());
    hd->twiddle = upload(tw.data(), tw.size());
    {
        std::vector<float> zz(64, 0.f);
        hd->zeros = upload(zz.data(), zz.size());
    }
    if (!hd->poly || !hd->twiddle || !hd->zeros) return bail(fail(MBX_ERR_HIP, "hipMemcpy of a table failed"));
    for (int u : ups) {
        if (hd->lerp.count(u)) continue;
        std::vector<float> w0(u), w1(u);
        for (int j = 0; j < u; ++j) {   // float32 of the float64 ratios (reference support_layers.py:19-27)
            w0[j] = (float)((double)(u - j) / (double)u);
            w1[j] = (float)((double)j / (double)u);
        }
        float *d0 = upload(w0.data(), u), *d1 = upload(w1.data(), u);
        if (!d0 || !d1) return bail(fail(MBX_ERR_HIP, "hipMemcpy of a table failed"));
        hd->lerp[u] = {d0, d1};
    }

    // required tensors
    std::vector<std::string> need = {"table.hann", "table.inv_win", "table.wavetables", "wn.start.w", "wn.start.b",
                                     "wn.end.w", "wn.end.b", "post.w", "post.b"};
    // conditioning chain (reference custom_AE_layers.py:190-227,283-289): pre-conditioning convolutions, then the
    // conditioning layer; all with kernel size cond_kernel_size and zero SAME padding, no activation in between
    auto cond_chain = [&](const std::string &prefix, int channels, int ccu, std::vector<mbx_subnet_op> &ops) {
        if (c.wn_disable_conditioning) return;
        int chan = c.mel_channels;
        auto add = [&](const std::string &nm, int cout) {
            mbx_subnet_op op{};
            op.kind = MBX_OP_CONV;
            op.ks = c.cond_kernel_size;
            op.cin = chan;
            op.cout = cout;
            op.pad_l = c.wn_causal ? c.cond_kernel_size - 1 : (c.cond_kernel_size - 1) / 2;
            op.pad_r = c.cond_kernel_size - 1 - op.pad_l;
            op.pad_mode = MBX_PAD_ZERO;
            op.up = 1;
            std::snprintf(op.name, MBX_NAME_LEN, "%s", nm.c_str());
            ops.push_back(op);
            need.push_back(nm + ".w");
            need.push_back(nm + ".b");
            chan = cout;
        };
        for (int i = 0; i < c.n_precond; ++i) {
            add(prefix + "precond_" + std::to_string(i), c.precond_channels[i]);
            hd->cond_buf_per_frame = std::max<long long>(hd->cond_buf_per_frame, c.precond_channels[i]);
        }
        add(prefix + "cond", 2 * channels * ccu);
    };
    cond_chain("wn.", c.wn_channels, c.cond_conv_upsampling, hd->cond_ops);
    // several WaveNet blocks (reference custom_pulsed_generator.py:456-488): every block has its own start / conditioning
    // / layer / end tensors; the up-sampling convolution "up<b>" sits behind block b
    if (c.n_wn_blocks >= 1) {
        int spf = spf0;
        for (int b = 0; b < c.n_wn_blocks; ++b) {
            mbx_handle::WnBlock blk;
            blk.C = c.wn_block_channels[b];
            blk.ups = c.wn_block_ups[b];
            blk.spf = spf;
            if (spf % c.cond_lin_upsampling) return bail(fail(MBX_ERR_INVALID_ARGUMENT, "a WaveNet block's rate is not a multiple of cond_lin_upsampling"));
            blk.ccu = spf / c.cond_lin_upsampling;
            blk.prefix = b == 0 ? "wn." : "wn" + std::to_string(b) + ".";
            if (b == 0) blk.cond_ops = hd->cond_ops;
            else cond_chain(blk.prefix, blk.C, blk.ccu, blk.cond_ops);
            hd->mb_hc_per_frame = std::max<long long>(hd->mb_hc_per_frame, (long long)spf * blk.C);
            if (b >= 1) {
                need.push_back(blk.prefix + "start.w");
                need.push_back(blk.prefix + "start.b");
                need.push_back(blk.prefix + "end.w");
                need.push_back(blk.prefix + "end.b");
                for (int l = 0; l < c.wn_layers; ++l)
                    for (const char *nm : {"conv1D_", "res_skip_"}) {
                        need.push_back(blk.prefix + nm + std::to_string(l) + ".w");
                        need.push_back(blk.prefix + nm + std::to_string(l) + ".b");
                    }
            }
            if (blk.ups > 1) {
                need.push_back("up" + std::to_string(b) + ".w");
                need.push_back("up" + std::to_string(b) + ".b");
            }
            spf *= blk.ups;
            hd->blocks.push_back(blk);
        }
    }
    if (c.nm_iters > 0) {
        if (c.nm_smooth_win < c.hop_size || c.nm_smooth_win % 2 || !(c.nm_rms_norm_fact > 0.f))
            return bail(fail(MBX_ERR_INVALID_ARGUMENT, "RMS normalisation: bad smoothing window / norm factor"));
        need.push_back("table.nm_inv_enorm");
        need.push_back("table.nm_gwin");
        need.push_back("table.nm_smooth_win");
        if (c.nm_use_pinv) {
            if (c.mel_channels > 256 || !(c.nm_win_norm > 0.f))
                return bail(fail(MBX_ERR_INVALID_ARGUMENT, "normalize_use_pinv: at most 256 mel channels, nm_win_norm > 0"));
            need.push_back("table.nm_pinv");
        }
    }
    if (c.pulse_pqmf_taps > 0) need.push_back("table.pulse_ana");
    if (c.n_ceps_windows) {
        need.push_back("table.ceps_windows");
        need.push_back("table.ceps_log10f0");
        need.push_back("table.f0_smooth");
    }
    for (int l = 0; l < c.wn_layers; ++l) {
        need.push_back("wn.conv1D_" + std::to_string(l) + ".w");
        need.push_back("wn.conv1D_" + std::to_string(l) + ".b");
        need.push_back("wn.res_skip_" + std::to_string(l) + ".w");
        need.push_back("wn.res_skip_" + std::to_string(l) + ".b");
    }
    for (const auto &nm : need)
        if (!find(hd, nm)) return bail(fail(MBX_ERR_INVALID_ARGUMENT, "missing tensor " + nm));
    auto expect = [&](const std::string &nm, long long count) {
        const DevTensor *t = find(hd, nm);
        return t && t->count == count;
    };
    const int C = c.wn_channels;
    bool ok = expect("wn.start.w", (long long)c.wn_in_channels * C) &&
              expect("wn.end.w", (long long)C * c.wn_out_channels) && expect("post.w", (long long)c.wn_out_channels * M) &&
              expect("table.hann", c.stft_win) && expect("table.inv_win", c.stft_win) &&
              expect("table.wavetables", (long long)(c.wt_n_period + 1) * c.wt_n_tables);
    for (const mbx_subnet_op &op : hd->cond_ops)
        ok = ok && expect(std::string(op.name) + ".w", (long long)op.ks * op.cin * op.cout) && expect(std::string(op.name) + ".b", op.cout);
    for (size_t b = 0; b < hd->blocks.size(); ++b) {
        const auto &blk = hd->blocks[b];
        const long long Cb = blk.C;
        if (b >= 1) {
            for (const mbx_subnet_op &op : blk.cond_ops)
                ok = ok && expect(std::string(op.name) + ".w", (long long)op.ks * op.cin * op.cout) && expect(std::string(op.name) + ".b", op.cout);
            ok = ok && expect(blk.prefix + "start.w", (long long)c.wn_out_channels * Cb) && expect(blk.prefix + "end.w", Cb * c.wn_out_channels);
            for (int l = 0; l < c.wn_layers && ok; ++l)
                ok = expect(blk.prefix + "conv1D_" + std::to_string(l) + ".w", (long long)c.wn_kernel_size * Cb * 2 * Cb) &&
                     expect(blk.prefix + "res_skip_" + std::to_string(l) + ".w", Cb * (l < c.wn_layers - 1 ? 2 * Cb : Cb));
        }
        if (blk.ups > 1)
            ok = ok && expect("up" + std::to_string(b) + ".w", 3LL * c.wn_out_channels * c.wn_out_channels * blk.ups);
    }
    for (int l = 0; l < c.wn_layers && ok; ++l) {
        ok = expect("wn.conv1D_" + std::to_string(l) + ".w", (long long)c.wn_kernel_size * C * 2 * C) &&
             expect("wn.res_skip_" + std::to_string(l) + ".w", (long long)C * (l < c.wn_layers - 1 ? 2 * C : C));
    }
    if (c.pulse_pqmf_taps > 0) ok = ok && expect("table.pulse_ana", (long long)(c.pulse_pqmf_taps + 1) * c.pulse_channels);
    if (c.n_ceps_windows)
        ok = ok && expect("table.ceps_windows", (long long)c.n_ceps_windows * c.n_ceps) &&
             expect("table.f0_smooth", 2 * c.hop_size + 1);
    if (c.nm_iters > 0)
        ok = ok && expect("table.nm_inv_enorm", c.mel_channels) && expect("table.nm_gwin", c.stft_win) &&
             expect("table.nm_smooth_win", c.nm_smooth_win) &&
             (!c.nm_use_pinv || expect("table.nm_pinv", (long long)c.mel_channels * (c.fft_size / 2 + 1)));
    if (!ok) return bail(fail(MBX_ERR_INVALID_ARGUMENT, "a tensor has the wrong number of elements"));

    long long pf0 = 0, pvtf = 0;
    int f0_out = 0, vtf_out = 0;
    mbx_status st = analyse_subnet(c.f0_ops, c.n_f0_ops, c.mel_channels, &pf0, &hd->f0_time_factor, &f0_out);
    if (st != MBX_OK) return bail(st);
    if (!c.ps_off) {
        st = analyse_subnet(c.vtf_ops, c.n_vtf_ops, c.mel_channels, &pvtf, &hd->vtf_time_factor, &vtf_out);
        if (st != MBX_OK) return bail(st);
    }
    if (hd->f0_time_factor < c.pulse_per_frame || f0_out != 1)
        return bail(fail(MBX_ERR_INVALID_ARGUMENT, "F0 sub-net must end with 1 channel at >= pulse_per_frame samples per frame"));
    if (!c.ps_off && (hd->vtf_time_factor != 1 || vtf_out != c.n_ceps))
        return bail(fail(MBX_ERR_INVALID_ARGUMENT, "VTF sub-net must end with n_ceps channels at the mel frame rate"));
    hd->subnet_buf_per_frame = std::max(pf0, pvtf);
    if (c.f0_accumulate != MBX_F0_ACC_F64 && c.f0_accumulate != MBX_F0_ACC_F32)
        return bail(fail(MBX_ERR_INVALID_ARGUMENT, "f0_accumulate must be MBX_F0_ACC_F64 or MBX_F0_ACC_F32"));
    hd->f0_full64 = f0_chain_is_full64(hd);
    if (c.wn_conv_form < MBX_CONV_AUTO || c.wn_conv_form > MBX_CONV_F43)
        return bail(fail(MBX_ERR_INVALID_ARGUMENT, "wn_conv_form must be MBX_CONV_AUTO, _DIRECT, _F23 or _F43"));
    if (c.tune_gate_shape < 0 || c.tune_gate_shape > 3 || c.tune_resskip_split < 0 || c.tune_resskip_split > 3 ||
        c.tune_resskip_wave_tiles < -1 || c.calib_fraction < 0.f || c.calib_fraction > 1.f)
        return bail(fail(MBX_ERR_INVALID_ARGUMENT, "tune_* / calib_fraction out of range"));
    {
        // skip path folded into the end convolution when the host supplied the folded tensors (wn_keep_skip: keep
        // the skip tensor, e.g. to look at the "wn_skip" stage)
        bool have = !c.wn_keep_skip && c.wn_out_channels <= 32 && M <= 16;
        const long long nct = (C + c.wn_out_channels + 127) / 128, nk = (C + 15) / 16;
        for (int l = 0; l + 1 < c.wn_layers && have; ++l)
            have = expect("wn.res_skip_" + std::to_string(l) + ".fold", nct * nk * 2048) &&
                   expect("wn.res_skip_" + std::to_string(l) + ".fold_b", C + c.wn_out_channels);
        have = have && expect("wn.tail.fold", (long long)((C + 7) / 8) * 256) && expect("wn.tail.fold_b", c.wn_out_channels);
        if (c.n_wn_blocks >= 1) have = false;      // several blocks: generic kernels (run_wavenet_blocks)
        const bool sym = !c.wn_causal;              // the folded first layer and the Winograd forms assume SAME padding
        hd->fold_skip = have;
        // start convolution folded into layer 0 (wn_gate0.hip); wn_keep_start keeps the h0 tensor and the full layer
        bool have0 = have && sym && !c.wn_keep_start && c.wn_kernel_size == 3 &&
                     mbx::wn_gate0_fits(C, c.pulse_channels * (1 + c.wt_subharm_channels), c.wn_dilations[0], c.cond_lin_upsampling) &&
                     expect("wn.conv1D_0.start_fold", (long long)((C + 31) / 32) * 1536);
        if (have0 && c.wn_layers > 1)
            have0 = expect("wn.res_skip_0.fold_start", nct * ((C + 16 + 15) / 16) * 2048);
        hd->fold_start = have0;
    }
    if (c.wn_precision != MBX_PRECISION_F32 && c.wn_precision != MBX_PRECISION_SPLIT_F16)
        return bail(fail(MBX_ERR_INVALID_ARGUMENT, "wn_precision must be MBX_PRECISION_F32 or MBX_PRECISION_SPLIT_F16"));
    if (c.wn_precision == MBX_PRECISION_SPLIT_F16) {
        // opt-in experiment: folded res/skip layers 1 .. L-2 on the 16-bit matrix pipe (wn_resskip_f16.hip)
        if (c.wn_gate_activation == MBX_GATE_GLU)
            return bail(fail(MBX_ERR_UNSUPPORTED, "wn_precision = split f16 needs a bounded gate (not glu)"));
        bool have16 = hd->fold_skip && c.wn_layers >= 3 && C + c.wn_out_channels <= 384;
        for (int l = 1; l + 1 < c.wn_layers && have16; ++l)
            have16 = expect("wn.res_skip_" + std::to_string(l) + ".fold_f16", (long long)((C + 31) / 32) * 12 * 1024);
        if (!have16)
            return bail(fail(MBX_ERR_INVALID_ARGUMENT, "wn_precision = split f16 needs the folded skip path, >= 3 layers, C + n_out <= 384 "
                                                        "and the wn.res_skip_<l>.fold_f16 images"));
        hd->split_f16 = true;
        // ... and the gate layers behind the folded first one (wn_gate_f16.hip), where the host supplied their images
        hd->split_f16_gate = hd->fold_start && !c.wn_causal && c.wn_kernel_size == 3;
        for (int l = 1; l < c.wn_layers && hd->split_f16_gate; ++l)
            hd->split_f16_gate = expect("wn.conv1D_" + std::to_string(l) + ".gate_f16", (long long)((C + 31) / 32) * ((C + 31) / 32) * 6144);
    }
    {
        hd->gate_small_shape = c.tune_gate_shape - 1;
        if (c.tune_resskip_wave_tiles) hd->resskip_wave_tiles = std::max(0, c.tune_resskip_wave_tiles);
        hd->resskip_split = c.tune_resskip_split;
        // form of the dilated convolution: a Winograd form needs its weight images (for every layer that runs the gate
        // kernels), SAME padding and kernel size 3; a handle without them runs the direct form whatever was asked for
        const bool can43 = form_available(hd, MBX_CONV_F43), can23 = form_available(hd, MBX_CONV_F23);
        int form = c.wn_conv_form;
        const bool autoform = form == MBX_CONV_AUTO;
        if (autoform) form = can43 ? MBX_CONV_F43 : can23 ? MBX_CONV_F23 : MBX_CONV_DIRECT;
        if (form == MBX_CONV_F43 && !can43) form = can23 ? MBX_CONV_F23 : MBX_CONV_DIRECT;
        if (form == MBX_CONV_F23 && !can23) form = MBX_CONV_DIRECT;
        set_form(hd, form);
        if ((autoform && form != MBX_CONV_DIRECT) || hd->split_f16) {
            // MBX_CONV_AUTO: the Winograd forms must earn their place on this handle's own weights -- and so must the opt-in
            // split precision, whatever the form (an overflow of fp16's range by the hidden state shows here as well)
            st = calibrate_on_synthetic_mel(hd, autoform && form != MBX_CONV_DIRECT);
            if (st != MBX_OK) return bail(st);
        }
    }
    *out = hd;
    return MBX_OK;
}

mbx_status mbx_destroy(mbx_handle *handle) {
    if (!handle) return MBX_OK;
    if (handle->arena) (void)hipFree(handle->arena);
    for (auto &pool : handle->ev_pool)
        for (auto &pr : pool) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    delete handle;
    return MBX_OK;
}

size_t mbx_workspace_size(const mbx_handle *handle, int32_t batch, int32_t max_frames) {
    if (!handle || batch <= 0 || max_frames <= 0) return 0;
    return carve(handle, nullptr, batch, max_frames).total;
}

// Geometry of the per-layer state a stream carries between ticks (mbx_forward_options.layer_store): layer l reaches
// r[l] rows to either side; it is exact up to row e_l = E - reach_rows + c[l] when the region ends at row E, with
// e_l = e_{l-1} - step[l] (step = the reach rounded up to even rows: the rows of the n_out-wide accumulator stay 8-byte
// aligned).  A slot keeps per layer l >= 1 the rows [e_l - r[l], e_l + step[l]) of h_l and [e_l, e_l + step[l]) of the
// accumulator.
struct LayerGeom {
    int floats, reach_rows, min_rows;
    int r[MBX_MAX_WN_LAYERS], step[MBX_MAX_WN_LAYERS], c[MBX_MAX_WN_LAYERS];
    long long off[MBX_MAX_WN_LAYERS];
};

static LayerGeom layer_geom(const mbx_handle *hd) {
    LayerGeom g{};
    const mbx_config &c = hd->cfg;
    const int L = c.wn_layers, C = c.wn_channels;
    // needs the folded graph (no separate start / skip tensors to carry) and the F(2,3) gate kernel (per-layer regions
    // start between conditioning rows: ConvArgs::cond_phase)
    if (!hd->fold_skip || !hd->fold_start || !hd->winograd || c.wn_kernel_size != 3 || L < 2) return g;
    for (int l = 1; l < L; ++l) {
        const DevTensor *wino = find(hd, "wn.conv1D_" + std::to_string(l) + ".wino2w");
        const int d = c.wn_dilations[l];
        if (!wino || d > 16 || (d & (d - 1)) != 0) return g;
    }
    for (int l = 0; l < L; ++l) {
        g.r[l] = c.wn_dilations[l] * (c.wn_kernel_size - 1) / 2;
        g.step[l] = (g.r[l] + 1) & ~1;
    }
    g.c[L - 1] = 0;
    for (int l = L - 2; l >= 0; --l) g.c[l] = g.c[l + 1] + g.step[l + 1];
    const int spf = c.steps_per_frame;
    // (+ cond_lin_upsampling - 1: a whole-region run interpolates the conditioning of its last rows towards the clamped
    // last conditioning row; that error spreads backwards through the layers behind, streaming.py::stream_margins)
    g.reach_rows = (g.c[0] + g.r[0] + c.cond_lin_upsampling - 1 + spf - 1) / spf * spf;
    long long off = 0;
    for (int l = 1; l < L; ++l) {
        g.off[l] = off;
        off += (long long)(g.r[l] + g.step[l]) * C + (long long)g.step[l] * c.wn_out_channels;
        g.min_rows = std::max(g.min_rows, g.r[l] + g.step[l]);
    }
    g.floats = (int)off;
    return g;
}

mbx_status mbx_layer_state_info(const mbx_handle *hd, int32_t *floats_per_slot, int32_t *reach_rows, int32_t *min_rows) {
    if (!hd) return fail(MBX_ERR_INVALID_ARGUMENT, "null handle");
    const LayerGeom g = layer_geom(hd);
    if (floats_per_slot) *floats_per_slot = g.floats;
    if (reach_rows) *reach_rows = g.reach_rows;
    if (min_rows) *min_rows = g.min_rows;
    return MBX_OK;
}

struct LayerOpts {
    float *store;
    int floats;
    const int32_t *carry;
    int rows;
};

// everything mbx_forward_stream / mbx_forward_ex add to mbx_forward (see mbx_forward_options in mbexwn.h)
struct ForwardExtras {
    const mbx::StreamState *st_in = nullptr;
    mbx::StreamState *st_out = nullptr;
    const float *f0_in = nullptr;
    float transposition = 1.f;
    int active_begin = 0;
    const int32_t *active_frames = nullptr;
    int wn_begin = 0;
    const int32_t *wn_frames = nullptr;
    float *sub_store = nullptr;
    int sub_store_rows = 0;
    const int32_t *sub_carry = nullptr;
    int active_max_frames = 0, wn_max_frames = 0;
    const LayerOpts *lay = nullptr;
    float *fe_store = nullptr;
    int fe_ring_frames = 0, fe_new_frames = 0, fe_margin_frames = 0, fe_end_frames = 0;
    const int32_t *fe_pos = nullptr;
};

// Several WaveNet blocks with in-block upsampling (reference custom_pulsed_generator.py:456-488, 908-914;
// custom_AE_layers.py:273-346, 574-582), on the generic kernels: block b = start convolution (block 0: fold + noise
// channel + start, wn_start_kernel) -> L x (dilated convolution + conditioning + gate, res/skip) -> end convolution ->
// sub-pixel convolution "up<b>" (depth -> time).  The last block's output goes through the post-net into the sub-band
// rows.  Whole items only (no stream regions).
static mbx_status run_wavenet_blocks(mbx_handle *hd, const Workspace &w, int B, int T, const int32_t *n_frames,
                                     const float *noise, hipStream_t stream) {
    const mbx_config &c = hd->cfg;
    const int L = c.wn_layers, n_out = c.wn_out_channels, M = c.subbands, cond_up = c.cond_lin_upsampling;
    const long long npulse = (long long)T * c.pulse_per_frame;
    const int nsub1 = 1 + c.wt_subharm_channels;
    auto lerp = hd->lerp[cond_up];
    const float *x_in = nullptr;                 // output of the previous block (B, rows, n_out)
    for (size_t b = 0; b < hd->blocks.size(); ++b) {
        const auto &blk = hd->blocks[b];
        const int C = blk.C, spf = blk.spf;
        const long long rows = (long long)T * spf;
        const bool last_block = b + 1 == hd->blocks.size();
        const float *cond = b == 0 ? w.cond : w.mb_cond[b];
        const long long cond_bstride = (long long)T * blk.ccu * 2 * C;
        const DevTensor *ws = find(hd, blk.prefix + "start.w"), *bs = find(hd, blk.prefix + "start.b");
        if (b == 0) {
            ScopedEvents ev(hd, PROF_START, stream);
            mbx::launch_wn_start(w.pulse, npulse * nsub1, c.noise_sigma != 0.f ? noise : nullptr, rows, c.noise_sigma, n_frames, spf,
                                 (int)rows, B, c.pulse_channels * nsub1, ws->ptr, bs->ptr, C, w.mb_h, rows * C, stream);
        } else {
            mbx::ConvArgs a = conv_args(x_in, rows * n_out, n_out, n_frames, spf, (int)rows, B, ws, bs, 1, n_out, C, 1, 0,
                                        MBX_PAD_ZERO, w.mb_h, rows * C, C);
            mbx::launch_conv1d(a, mbx::EPI_LINEAR, stream);
        }
        for (int l = 0; l < L; ++l) {
            const std::string ls = std::to_string(l);
            const int d = c.wn_dilations[l];
            mbx::ConvArgs g = conv_args(w.mb_h, rows * C, C, n_frames, spf, (int)rows, B, find(hd, blk.prefix + "conv1D_" + ls + ".w"),
                                        find(hd, blk.prefix + "conv1D_" + ls + ".b"), c.wn_kernel_size, C, 2 * C, d,
                                        (c.wn_causal ? d * (c.wn_kernel_size - 1) : d * (c.wn_kernel_size - 1) / 2), MBX_PAD_ZERO, w.mb_a, rows * C, C);
            g.cond = cond;
            g.cond_bstride = cond_bstride;
            g.cond_up = cond_up;
            g.lerp_w0 = lerp.first;
            g.lerp_w1 = lerp.second;
            g.channels = C;
            g.gate_act = c.wn_gate_activation;
            g.zeros = hd->zeros;
            {
                // the Winograd F(4,3) form when the host supplied the block's weight image and the layer fits (SAME padding,
                // k = 3, power-of-two dilation <= 16), the direct form otherwise
                ScopedEvents ev(hd, PROF_GATE, stream);
                bool done = false;
                const DevTensor *wino4 = (hd->winograd == 4 && !c.wn_causal) ? find(hd, blk.prefix + "conv1D_" + ls + ".wino4w") : nullptr;
                if (wino4 && wino4->ndim == 3 && wino4->shape[0] == (C + 31) / 32 && wino4->shape[1] == (C + 7) / 8 &&
                    wino4->shape[2] == 3072) {
                    mbx::ConvArgs gw = g;
                    gw.w = wino4->ptr;
                    done = mbx::launch_wn_gate_winograd4w(gw, 0, stream);
                }
                if (!done) mbx::launch_conv1d(g, mbx::EPI_GATE, stream);
            }
            const bool last = l == L - 1;
            mbx::ConvArgs r = conv_args(w.mb_a, rows * C, C, n_frames, spf, (int)rows, B, find(hd, blk.prefix + "res_skip_" + ls + ".w"),
                                        find(hd, blk.prefix + "res_skip_" + ls + ".b"), 1, C, last ? C : 2 * C, 1, 0, MBX_PAD_ZERO,
                                        nullptr, 0, 0);
            r.channels = C;
            r.zeros = hd->zeros;
            r.h = w.mb_h;
            r.skip = w.mb_skip;
            r.hs_bstride = rows * C;
            r.skip_init = (l == 0);
            r.last_layer = last;
            ScopedEvents ev(hd, PROF_RES_SKIP, stream);
            const DevTensor *pk = find(hd, blk.prefix + "res_skip_" + ls + ".packed");
            const int cout_l = last ? C : 2 * C;
            bool rdone = false;
            if (pk && pk->ndim == 3 && pk->shape[0] == (cout_l + 127) / 128 && pk->shape[1] == (C + 15) / 16 && pk->shape[2] == 2048) {
                mbx::ConvArgs rp = r;
                rp.w = pk->ptr;
                rdone = mbx::launch_wn_resskip(rp, stream);
            }
            if (!rdone) mbx::launch_conv1d(r, mbx::EPI_RESSKIP, stream);
        }
        // end convolution (reference custom_AE_layers.py:337-340); the last block's output is the stage "wn_out" unless an
        // up-sampling convolution follows it
        ScopedEvents ev(hd, PROF_TAIL, stream);
        float *y = (last_block && blk.ups == 1) ? w.wn_out : w.mb_y0;
        mbx::ConvArgs e = conv_args(w.mb_skip, rows * C, C, n_frames, spf, (int)rows, B, find(hd, blk.prefix + "end.w"),
                                    find(hd, blk.prefix + "end.b"), 1, C, n_out, 1, 0, MBX_PAD_ZERO, y, rows * n_out, n_out);
        mbx::launch_conv1d(e, mbx::EPI_LINEAR, stream);
        x_in = y;
        if (blk.ups > 1) {
            // Conv1DUpDownSample (reference conv_layers.py:177-261): k = 3 convolution to n_out * ups channels, zero SAME
            // padding, then depth -> time: row r of the input becomes rows ups r .. ups r + ups - 1 (a reshape in memory)
            const std::string un = "up" + std::to_string(b);
            float *yu = last_block ? w.wn_out : w.mb_y1;
            mbx::ConvArgs u = conv_args(y, rows * n_out, n_out, n_frames, spf, (int)rows, B, find(hd, un + ".w"), find(hd, un + ".b"), 3,
                                        n_out, n_out * blk.ups, 1, c.wn_causal ? 2 : 1, MBX_PAD_ZERO, yu, rows * n_out * blk.ups, n_out * blk.ups);
            mbx::launch_conv1d(u, mbx::EPI_LINEAR, stream);
            x_in = yu;
        }
    }
    // post-net 1x1 (reference custom_pulsed_generator.py:490-493,913-914) at the sub-band rate
    const long long nsteps = (long long)T * c.steps_per_frame;
    mbx::ConvArgs pn = conv_args(x_in, nsteps * n_out, n_out, n_frames, c.steps_per_frame, (int)nsteps, B, find(hd, "post.w"),
                                 find(hd, "post.b"), 1, n_out, M, 1, 0, MBX_PAD_ZERO, w.sub, nsteps * M, M);
    mbx::launch_conv1d(pn, mbx::EPI_LINEAR, stream);
    return MBX_OK;
}

static mbx_status forward_impl(mbx_handle *hd, const float *mel, const int32_t *n_frames, int32_t batch,
                               int32_t max_frames, const float *noise, float *audio, void *workspace,
                               size_t workspace_bytes, void *hip_stream, const ForwardExtras &ex = ForwardExtras()) {
    const mbx::StreamState *st_in = ex.st_in;
    mbx::StreamState *st_out = ex.st_out;
    const float *f0_in = ex.f0_in;
    const float transposition = ex.transposition;
    const int active_begin = ex.active_begin, wn_begin = ex.wn_begin, sub_store_rows = ex.sub_store_rows;
    const int32_t *active_frames = ex.active_frames, *wn_frames = ex.wn_frames, *sub_carry = ex.sub_carry;
    float *sub_store = ex.sub_store;
    const int active_max_frames = ex.active_max_frames, wn_max_frames = ex.wn_max_frames;
    const LayerOpts *lay = ex.lay;
    if (!hd || !mel || !audio || !workspace) return fail(MBX_ERR_INVALID_ARGUMENT, "null argument");
    // mel-rate front end carried between the ticks of a stream (mbx_forward_options.fe_store): fe_frames > 0 = only the
    // last fe_frames frames of the window are computed, the frames in front of them come from the ring
    const bool fe_on = ex.fe_store != nullptr;
    int fe_frames = 0;
    const int fe_end = ex.fe_end_frames > 0 ? ex.fe_end_frames : max_frames;      // frames of the items of a steady tick
    if (fe_on) {
        if (!ex.fe_pos || !ex.sub_carry || ex.fe_ring_frames < max_frames || ex.fe_new_frames < 0 || ex.fe_margin_frames < 0 ||
            fe_end > max_frames || ex.fe_new_frames + ex.fe_margin_frames > fe_end)
            return fail(MBX_ERR_INVALID_ARGUMENT, "fe_store needs fe_pos, sub_carry (slots), fe_ring_frames >= max_frames, "
                                                  "fe_end_frames <= max_frames and fe_new_frames + fe_margin_frames <= fe_end_frames");
        if (hd->cfg.nm_iters > 0 || hd->f0_time_factor > hd->cfg.pulse_per_frame || ex.f0_in)
            return fail(MBX_ERR_UNSUPPORTED, "the front end cannot be carried for this model / call (RMS normalisation, an F0-net "
                                             "that runs above the pulse rate, an external F0 contour)");
        if ((2 * hd->cfg.wn_channels * hd->cfg.cond_conv_upsampling) % 4 || hd->cfg.n_ceps % 4 || hd->cfg.pulse_per_frame % 4)
            return fail(MBX_ERR_UNSUPPORTED, "the front end ring needs per-frame sizes that are multiples of 4 floats");
        if (ex.fe_new_frames > 0) fe_frames = ex.fe_new_frames + ex.fe_margin_frames;
    }
    if (batch <= 0 || max_frames <= 0) return fail(MBX_ERR_INVALID_ARGUMENT, "batch and max_frames must be positive");
    // row and sample indices inside the kernels are 32-bit ints (offsets are 64-bit or block relative): 2^24 sub-band
    // rows per item = 2.9 h of audio at the canonical 1.6 kHz is the tested side of that
    if ((long long)max_frames * hd->cfg.steps_per_frame >= (1LL << 24))
        return fail(MBX_ERR_UNSUPPORTED, "an item may have at most 2^24 - 1 sub-band rows (split longer recordings)");
    if ((!hd->blocks.empty() || hd->cfg.wn_causal) && (active_frames || st_in || st_out || sub_carry || ex.lay || fe_on))
        return fail(MBX_ERR_UNSUPPORTED, "a model with several WaveNet blocks or causal padding runs whole items only (no stream windows / state)");
    DeviceGuard guard(hd->device);
    if (!guard.ok) return fail(MBX_ERR_HIP, "cannot select the handle's device");
    const mbx_config &c = hd->cfg;
    if (c.noise_sigma != 0.f && !noise) return fail(MBX_ERR_INVALID_ARGUMENT, "noise is required when noise_sigma != 0");
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(MBX_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    const int B = batch, T = max_frames;
    if (active_begin < 0 || active_begin >= T || (active_begin > 0 && !active_frames))
        return fail(MBX_ERR_INVALID_ARGUMENT, "active_begin must lie inside the window and needs active_frames");
    if (wn_frames && (!active_frames || wn_begin < active_begin || wn_begin >= T))
        return fail(MBX_ERR_INVALID_ARGUMENT, "wn_frames needs active_frames and wn_begin inside the active region");
    if ((sub_store != nullptr) != (sub_carry != nullptr) || (sub_store && sub_store_rows <= 0))
        return fail(MBX_ERR_INVALID_ARGUMENT, "sub_store, sub_store_rows and sub_carry go together");
    Workspace w = carve(hd, static_cast<char *>(workspace), B, T);
    if (w.total > workspace_bytes) return fail(MBX_ERR_WORKSPACE, "workspace too small, see mbx_workspace_size");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const int C = c.wn_channels, L = c.wn_layers, M = c.subbands;
    const long long npulse = (long long)T * c.pulse_per_frame, nsteps = (long long)T * c.steps_per_frame;

    // ---- optional RMS normalisation of the mel input (reference wavegen_1d.py:493-495, 638-769); the matching gain
    // is multiplied onto the audio at the end (reference wavegen_1d.py:506-507)
    const float *nm_gain_src = nullptr;
    if (c.nm_iters > 0) {
        // (streaming windows: the smoothing treats the window edges as item edges, so the frames within the smoothing's
        // reach of a window edge differ from the whole-utterance run -- the caller's margins cover that reach,
        // streaming.py::norm_reach)
        ScopedEvents ev(hd, PROF_NORM_MEL, stream);
        nm_gain_src = mbx::launch_norm_mel(norm_mel_consts(hd), mel, (long long)T * c.mel_channels, n_frames, T, B,
                                           w.nm_a, w.nm_b, w.mel_norm, stream);
        mel = w.mel_norm;
    }
    // the float64 F0 chain reads the mel rows as 16-byte pieces: a caller's mel that starts in the middle of one (a view
    // into a larger buffer) is copied into the workspace first, so that the same handle and the same mel give the same
    // bits however the call was made
    if (hd->f0_full64 && c.nm_iters == 0 && (reinterpret_cast<uintptr_t>(mel) & 15) != 0 && c.mel_channels % 4 == 0) {
        if (hipMemcpyAsync(w.mel_norm, mel, (size_t)B * T * c.mel_channels * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return fail(MBX_ERR_HIP, "hipMemcpyAsync of a misaligned mel failed");
        mel = w.mel_norm;
    }
    const int cond_cout = 2 * C * c.cond_conv_upsampling;
    mbx::StftConsts sc = stft_consts(hd);
    // ---- conditioning conv (reference custom_AE_layers.py:214-227,287), VTF-net -> cepstrum (reference
    // custom_pulsed_generator.py:793-800) and F0-net (reference :773-791) are independent chains on the mel input:
    // the n-th convolution of each goes into one launch (launch_conv1d_group; matters at small batch, where the
    // mel-rate convolutions are latency-bound)
    {
        ScopedEvents ev(hd, PROF_FRONTEND, stream);
        // a carried front end computes the frames [fe_end - fe_frames, fe_end) of every item only (all items then have
        // fe_end frames: a steady tick); batch strides stay those of the window
        const int Tf = fe_frames ? fe_frames : T;
        const long long f_off = fe_frames ? fe_end - Tf : 0;
        const int32_t *nf_fe = fe_frames ? nullptr : n_frames;
        const float *mel_fe = mel + f_off * c.mel_channels;
        SubnetRun cond(hd, hd->cond_ops.data(), (int)hd->cond_ops.size(), mel_fe, c.mel_channels, nf_fe, B, Tf, w.sub4,
                       w.sub5, w.cond + f_off * cond_cout, false, 1.f, 0.f, stream);
        if (hd->cond_ops.empty()) {   // disable_conditioning: cond_layers = zeros (reference custom_AE_layers.py:293-294)
            cond.finished = true;
            if (hipMemsetAsync(w.cond, 0, (size_t)B * T * cond_cout * sizeof(float), stream) != hipSuccess)
                return fail(MBX_ERR_HIP, "hipMemsetAsync of the conditioning rows failed");
        }
        SubnetRun vtf(hd, c.vtf_ops, c.n_vtf_ops, mel_fe, c.mel_channels, nf_fe, B, Tf, w.sub2, w.sub3,
                      w.ceps + f_off * c.n_ceps, false, 1.f, 0.f, stream);
        const bool f0_wide = hd->f0_time_factor > c.pulse_per_frame;
        SubnetRun f0(hd, c.f0_ops, c.n_f0_ops, mel_fe, c.mel_channels, nf_fe, B, Tf, w.sub0, w.sub1,
                     f0_wide ? w.f0_wide : w.f0 + f_off * c.pulse_per_frame, true, c.f0_max - c.f0_min, c.f0_min, stream);
        f0.precise = c.f0_accumulate == MBX_F0_ACC_F64;
        // (a misaligned mel was copied into the workspace above; mel_channels % 4 != 0 keeps the float32 hidden layers)
        if (hd->f0_full64 && (reinterpret_cast<uintptr_t>(mel_fe) & 15) == 0 && c.mel_channels % 4 == 0) {
            f0.buf64_0 = w.f0h0;
            f0.buf64_1 = w.f0h1;
        }
        if (c.ps_off) vtf.finished = true;             // no VTF-net (the cepstrum buffer stays unused)
        if (fe_frames) {
            vtf.window_stride(T, c.mel_channels);
            f0.window_stride(T, c.mel_channels);
            cond.window_stride(T, c.mel_channels);
        }
        if (f0_in) f0.finished = true;
        for (;;) {
            mbx::ConvArgs group[3];
            int n = 0;
            if (f0.next_conv(group[n])) ++n;
            if (vtf.next_conv(group[n])) ++n;
            if (cond.next_conv(group[n])) ++n;
            if (!n) break;
            mbx::launch_conv1d_group(group, n, stream);
        }
        if (vtf.status != MBX_OK) return vtf.status;
        if (f0.status != MBX_OK) return f0.status;
        if (cond.status != MBX_OK) return cond.status;
        // the conditioning chains of the WaveNet blocks behind the first one (each block has its own layer)
        for (size_t bk = 1; bk < hd->blocks.size(); ++bk) {
            const auto &blk = hd->blocks[bk];
            const size_t floats = (size_t)B * T * blk.ccu * 2 * blk.C;
            if (blk.cond_ops.empty()) {
                if (hipMemsetAsync(w.mb_cond[bk], 0, floats * sizeof(float), stream) != hipSuccess)
                    return fail(MBX_ERR_HIP, "hipMemsetAsync of the conditioning rows failed");
                continue;
            }
            SubnetRun cb(hd, blk.cond_ops.data(), (int)blk.cond_ops.size(), mel, c.mel_channels, n_frames, B, T, w.sub4, w.sub5,
                         w.mb_cond[bk], false, 1.f, 0.f, stream);
            mbx::ConvArgs one;
            while (cb.next_conv(one)) mbx::launch_conv1d_group(&one, 1, stream);
            if (cb.status != MBX_OK) return cb.status;
        }
        if (f0_wide && !f0_in)   // pulse_frequency[:, :T * pulse_per_frame] (reference custom_pulsed_generator.py:787)
            mbx::launch_activation(w.f0_wide, (long long)T * hd->f0_time_factor, n_frames, c.pulse_per_frame, (int)npulse,
                                   B, 1, MBX_ACT_LINEAR, 1.f, 0.f, w.f0, npulse, stream);
    }
    if (f0_in)   // externally supplied contour (reference wavegen_1d.py:546-550)
        mbx::launch_activation(f0_in, npulse, n_frames, c.pulse_per_frame, (int)npulse, B, 1, MBX_ACT_LINEAR, transposition,
                               0.f, w.f0, npulse, stream);
    else if (transposition != 1.f)
        mbx::launch_activation(w.f0, npulse, n_frames, c.pulse_per_frame, (int)npulse, B, 1, MBX_ACT_LINEAR,
                               transposition, 0.f, w.f0, npulse, stream);
    if (fe_on) {
        // frames in front of the new ones come from the ring (they were computed, exactly, by earlier ticks); the new
        // ones (fe_new_frames == 0: the whole window) go there for the ticks to come
        ScopedEvents ev(hd, PROF_FRONTEND, stream);
        mbx::FrontendCarryArgs fc{};
        fc.cond = w.cond;
        fc.ceps = w.ceps;
        fc.f0 = w.f0;
        fc.cond_floats = cond_cout;
        fc.ceps_floats = c.n_ceps;
        fc.f0_floats = c.pulse_per_frame;
        fc.frames = T;
        fc.n_frames = n_frames;
        fc.store = ex.fe_store;
        fc.ring_frames = ex.fe_ring_frames;
        fc.pos = ex.fe_pos;
        fc.slot_desc = ex.sub_carry;
        fc.first_new = fe_frames ? fe_end - ex.fe_new_frames : 0;
        mbx::launch_frontend_carry(fc, B, stream);
    }
    // ---- wavetable excitation (reference :889)
    {
        ScopedEvents ev(hd, PROF_WAVETABLE, stream);
        mbx::launch_wavetable(wavetable_consts(hd), w.f0, npulse, n_frames, c.pulse_per_frame, (int)npulse, B, w.pulse,
                              nullptr, w.cum, w.chunk_last, st_in, st_out, stream);
    }
    // ---- PQMF analysis of the pulse signal instead of folding consecutive samples (reference :892-895)
    float *const pulse_osc = w.pulse;
    if (c.pulse_pqmf_taps > 0) {
        if (active_frames || st_in || st_out) return fail(MBX_ERR_UNSUPPORTED, "pulse_channels_use_pqmf models run whole items only");
        ScopedEvents ev(hd, PROF_WAVETABLE, stream);
        mbx::launch_pulse_analysis(w.pulse, npulse, n_frames, c.pulse_per_frame, (int)npulse, B, find(hd, "table.pulse_ana")->ptr,
                                   c.pulse_pqmf_taps, c.pulse_channels, w.pulse_ana, stream);
        w.pulse = w.pulse_ana;                   // what the WaveNet reads; the stage "pulse" stays the oscillator's output
    }
    // ---- active region (streaming windows, mbx_forward_options): from here on every stage sees the frames
    // [active_begin, active_begin + active_frames[b]) of the window as the item.  All buffers are (batch, frames * k)
    // with the batch stride of the whole window, so a region is a row offset into every buffer plus per-item row counts.
    const Workspace &w_base = w;                     // the stage table below points at the whole window
    const int32_t *n_frames_act = active_frames ? active_frames : n_frames;   // PQMF, STFT filter, overlap-add
    const long long act0 = active_frames ? active_begin : 0;
    // the WaveNet may have a region of its own inside the active one (wn_begin / wn_frames)
    const long long wn0 = active_frames ? (wn_frames ? wn_begin : active_begin) : 0;
    const int32_t *n_frames_wn = active_frames ? (wn_frames ? wn_frames : active_frames) : n_frames;
    // upper bounds of the rows an item can have in its region: the launchers size their grids (and pick block shapes)
    // from them; batch strides stay those of the whole window
    int wn_frames_max = T - (int)wn0;
    const int wn_bound = wn_frames ? wn_max_frames : active_max_frames;
    if (active_frames && wn_bound > 0) wn_frames_max = std::min(wn_frames_max, wn_bound);
    const int wn_rows = wn_frames_max * c.steps_per_frame;
    int act_frames_max = T - (int)act0;
    if (active_frames && active_max_frames > 0) act_frames_max = std::min(act_frames_max, active_max_frames);
    bool planes_only_run = false;      // set by single_block: split precision ran with the fp16 planes as the hidden state
    // ---- WaveNet (reference custom_AE_layers.py:273-346): one block (the measured path) or several (generic kernels)
    auto single_block = [&]() -> mbx_status {
    const bool fold_start = hd->fold_start;
    const bool fold = hd->fold_skip;
    const int n_out = c.wn_out_channels, spf = c.steps_per_frame, cond_up = c.cond_lin_upsampling;
    const int lda0 = (fold_start && L > 1) ? C + 16 : C;      // row stride of layer 0's output
    // A span = the rows of the window one launch treats as the item: first row, per-item row counts (nf[b] * rpf, or
    // max_rows for every item when nf is null), conditioning-rate phase of the first row.  Whole-region runs use one span
    // for every launch; a steady streaming tick (layer state carried, mbx_forward_options.layer_rows) one per layer.
    // out0 / out_rows: the rows of a gate span that are consumed (the others are the layer's reach: inputs only)
    struct Span { long long row0; const int32_t *nf; int rpf, max_rows, cphase, out0, out_rows; };
    const Span region{wn0 * spf, n_frames_wn, spf, wn_rows, 0, 0, 0};
    std::vector<Span> gate_sp(L, region), res_sp(L, region);
    Span tail_sp = region;
    const bool carry_layers = lay && lay->carry;
    const bool layered = carry_layers && lay->rows > 0;
    LayerGeom geo{};
    if (carry_layers) {
        geo = layer_geom(hd);
        if (!geo.floats) return fail(MBX_ERR_UNSUPPORTED, "this handle cannot carry layer state (mbx_layer_state_info)");
        if (!lay->store || lay->floats != geo.floats)
            return fail(MBX_ERR_INVALID_ARGUMENT, "layer_store / layer_store_floats do not match mbx_layer_state_info");
    }
    if (layered) {
        // every item: state stored up to row E - N, region end E; layer l runs on [s_l, s_l + N), s_l = E - N - reach + c_l
        const int N = lay->rows;
        if (!wn_frames || wn_max_frames <= 0 || N < geo.min_rows)
            return fail(MBX_ERR_INVALID_ARGUMENT, "layer_rows needs wn_frames, wn_max_frames and at least min_rows rows");
        const long long E = (long long)(wn_begin + wn_max_frames) * spf;
        if (E > nsteps || E - N - geo.reach_rows != (long long)wn_begin * spf)
            return fail(MBX_ERR_INVALID_ARGUMENT, "layer_rows: wn_begin must be the frame of the first new sub-band row");
        for (int l = 0; l < L; ++l) {
            const long long s = E - N - geo.reach_rows + geo.c[l], e = s + N;
            const int align = l == 0 ? cond_up : 2 * c.wn_dilations[l];      // Winograd F(2,3) pairs rows t, t+d in blocks of 2d
            const long long A = ((s - geo.r[l]) / align) * align;
            if (s - geo.r[l] < 0) return fail(MBX_ERR_INVALID_ARGUMENT, "layer_rows: the window does not reach far enough back");
            const int phase = (int)(A % cond_up);
            // rows of the launch: up to the layer's reach behind the last new row, and far enough that the conditioning
            // row behind the last new row is not the item's last one (which is where the interpolation clamps)
            const long long n1 = e + geo.r[l] - A;
            const long long n2 = (long long)cond_up * ((e - 1 - A + phase) / cond_up + 2) - phase;
            const long long need = std::max(n1, n2);
            if (A + need > nsteps) return fail(MBX_ERR_INVALID_ARGUMENT, "layer_rows: the region ends too close to the window end");
            // only the rows [s, e) are consumed (res_sp): the gate kernel computes the aligned blocks of output pairs that hold them
            const long long o0 = ((s - A) / align) * align;
            gate_sp[l] = Span{A, nullptr, 1, (int)need, phase, (int)o0, (int)(e - A - o0)};
            res_sp[l] = Span{s, nullptr, 1, N, 0, 0, 0};
        }
        tail_sp = res_sp[L - 1];
    }
    const int nsub1 = 1 + c.wt_subharm_channels;              // floats per excitation sample (pulse + sub-harmonic sinusoids)
    const int ppr = c.pulse_per_frame / spf * nsub1;           // excitation floats per WaveNet row
    if (c.pulse_per_frame % spf != 0) return fail(MBX_ERR_INVALID_ARGUMENT, "pulse_per_frame must be a multiple of steps_per_frame");
    if (!fold_start) {
        ScopedEvents ev(hd, PROF_START, stream);
        const Span &sp = region;
        mbx::launch_wn_start(w.pulse + sp.row0 * ppr, npulse * nsub1, c.noise_sigma != 0.f ? noise + sp.row0 : nullptr, nsteps, c.noise_sigma,
                             sp.nf, sp.rpf, sp.max_rows, B, c.pulse_channels * nsub1, find(hd, "wn.start.w")->ptr,
                             find(hd, "wn.start.b")->ptr, C, w.h + sp.row0 * C, nsteps * C, stream);
    }
    auto lerp = hd->lerp[cond_up];
    bool planes_valid = false;        // split half precision: the last res/skip launch also wrote h as fp16 planes (w.h16)
    // split half precision, round 6: where every consumer of the hidden state takes the planes -- the split gate of every layer
    // behind the first one (dilation <= 16: wn_gate_f16.hip) and the split res/skip layer of every layer in front of the last
    // one, layer 0's with the folded start convolution included -- the planes ARE the hidden state: hi + 2^-11 lo' keeps 22-23 of
    // float32's 24 bits, and the float32 tensor h (328 MB written and read again per layer at 16 x 10 s) is not touched
    bool planes_only = hd->split_f16 && hd->split_f16_gate && fold && fold_start && !st_in && !st_out && !active_frames && !lay && L >= 2 &&
                       find(hd, "wn.res_skip_0.fold_start_f16") != nullptr;
    for (int l = 1; l < L && planes_only; ++l) planes_only = c.wn_dilations[l] <= 16;
    for (int l = 1; l + 1 < L && planes_only; ++l) planes_only = find(hd, "wn.res_skip_" + std::to_string(l) + ".fold_f16") != nullptr;
    planes_only_run = planes_only;
    hd->last_gate_layers = L;
    for (int l = 0; l < L; ++l) {
        const std::string ls = std::to_string(l);
        const int d = c.wn_dilations[l];
        const Span &gs = gate_sp[l];
        mbx::ConvArgs g = conv_args(w.h + gs.row0 * C, nsteps * C, C, gs.nf, gs.rpf, gs.max_rows, B,
                                    find(hd, "wn.conv1D_" + ls + ".w"), find(hd, "wn.conv1D_" + ls + ".b"),
                                    c.wn_kernel_size, C, 2 * C, d, (c.wn_causal ? d * (c.wn_kernel_size - 1) : d * (c.wn_kernel_size - 1) / 2), MBX_PAD_ZERO, w.a + gs.row0 * C,
                                    nsteps * C, C);
        g.cond = w.cond + (gs.row0 / cond_up) * (2 * C);
        g.cond_bstride = (long long)T * cond_cout;
        g.cond_up = cond_up;
        g.cond_phase = gs.cphase;
        g.out_row0 = gs.out0;
        g.out_rows = gs.out_rows;
        g.lerp_w0 = lerp.first;
        g.lerp_w1 = lerp.second;
        g.channels = C;
        g.gate_act = c.wn_gate_activation;
        g.zeros = hd->zeros;
        if (l == 0 && fold_start) {
            // start convolution folded into the layer: a K = 24 contraction of the excitation (wn_gate0.hip)
            ScopedEvents ev(hd, PROF_GATE0, stream);
            mbx::Gate0Args g0{};
            g0.pulse = w.pulse + gs.row0 * ppr;
            g0.pulse_bstride = npulse * nsub1;
            g0.noise = c.noise_sigma != 0.f ? noise + gs.row0 : nullptr;
            g0.noise_bstride = nsteps;
            g0.sigma = c.noise_sigma;
            g0.pulse_channels = c.pulse_channels * nsub1;
            g0.n_frames = gs.nf;
            g0.rows_per_frame = gs.rpf;
            g0.max_rows = gs.max_rows;
            g0.batch = B;
            g0.w = find(hd, "wn.conv1D_0.start_fold")->ptr;
            g0.bias = g.bias;
            g0.channels = C;
            g0.gate_act = c.wn_gate_activation;
            g0.dil = d;
            g0.cond = g.cond;
            g0.cond_bstride = g.cond_bstride;
            g0.cond_up = g.cond_up;
            g0.lerp_w0 = g.lerp_w0;
            g0.lerp_w1 = g.lerp_w1;
            g0.out = w.a + gs.row0 * lda0;
            g0.out_bstride = nsteps * lda0;
            g0.ldo = lda0;
            g0.write_inputs = L > 1;
            if (gs.cphase != 0 || !mbx::launch_wn_gate0(g0, stream))
                return fail(MBX_ERR_INVALID_ARGUMENT, "folded first layer does not fit its kernel");
            hd->last_gate_kernel[l] = MBX_GATE_K_FOLDED_START;
        } else {
            ScopedEvents ev(hd, PROF_GATE, stream);
            // per-layer state of a stream: the rows this layer reads from in front of its own come from the item's slot,
            // the rows the next tick will read go there (layer_carry_kernel)
            if (carry_layers && l >= 1) {
                mbx::LayerCarryArgs lc{};
                lc.h = w.h;
                lc.h_bstride = nsteps * C;
                lc.C = C;
                lc.acc = w.wn_out;
                lc.acc_bstride = nsteps * n_out;
                lc.n_out = n_out;
                lc.store = lay->store;
                lc.slot_stride = geo.floats;
                lc.layer_off = geo.off[l];
                lc.desc = lay->carry;
                lc.inject = layered;
                lc.base_off = geo.c[l] - geo.reach_rows;
                lc.h_before = geo.r[l];
                lc.h_rows = geo.r[l] + geo.step[l];
                lc.acc_rows = geo.step[l];
                mbx::launch_layer_carry(lc, B, stream);
            }
            bool done = false;
            // F(4,3) block shape: what a launch of a few blocks per CU costs is the largest number of wave tiles a SIMD gets.
            // 256-row blocks put one whole tile (16 groups x 64 columns x 6 products) on every SIMD of their CU, the 128-row
            // product-split blocks half a tile: the finer shape runs when its worst SIMD gets clearly less work (a 10 s
            // utterance: 630 blocks = 2.46 per CU -> 3 tiles, against 1250 = 4.88 -> 5 halves).  Both give the same bits.
            // Streams run F(2,3): a window is bit-identical to an offline result only if both use one form with one group
            // alignment (streaming.py), and F(2,3) needs the shorter alignment; MBX_CONV_F23 makes offline runs use it too.
            const long long full_blocks = ((nsteps + 255) / 256) * B * ((C + 31) / 32);
            const long long half_blocks = ((nsteps + 127) / 128) * B * ((C + 31) / 32);
            const double load_full = (double)((full_blocks + 255) / 256), load_half = 0.5 * (double)((half_blocks + 255) / 256);
            // measured (scripts/experiments/gate_shapes.py, one item of 3 / 5 / 10 / 15 s: 51 / 65 / 102 / 155 us against 53 / 80 /
            // 110 / 167 us; two items of 10 s: 194 against 176 us): the finer shape wins up to about one resident round of
            // 256-row blocks and loses behind it, where both shapes divide evenly and the 128-row block's extra LDS-DMA traffic
            // per MFMA (the weight slice serves half the rows) and shorter slices tell
            bool split4 = !hd->winograd4_always && full_blocks <= 1024 && load_half <= load_full;
            // round 5: product-split blocks of HALF a column tile (one channel parity: twice the blocks of half the work; same bits
            // as the other shapes).  The idea: a 3 s utterance is 380 product-split blocks on 256 CUs, the worst CU works two
            // while the average is 1.5; 760 half blocks give every CU three.
            // Measured (scripts/experiments/gate_shapes.py, profiles/r05_gate_shapes.txt; one item of 1 / 2 / 3 / 4 / 5 / 10 s):
            // 32.4 / 43.8 / 54.0 / 65.9 / 70.4 / 131.8 us against 32.1 / 48.5 / 48.8 / 64.3 / 63.5 / 100.7 us for the product-split
            // blocks -- the estimate above was wrong: a block of half the matrix work lasts almost as long (40 barrier-separated
            // slices whose round trips, not whose MFMAs, set its time), so the finer shape only pays where the product-split
            // blocks leave CUs empty (<= 256 of them) and the half blocks do not (> 256): utterances around 2 s.
            int shape4 = split4 ? ((half_blocks <= 256 && 2 * half_blocks > 256) ? 2 : 1) : 0;
            if (hd->gate_small_shape >= 0 && !hd->winograd4_always && full_blocks < 4 * 768) shape4 = hd->gate_small_shape;
            split4 = shape4 != 0;
            // opt-in split half precision: whole-item forwards of the layers behind the folded first one
            if (hd->split_f16_gate && !st_in && !st_out && gs.cphase == 0 && gs.out_rows == 0) {
                mbx::ConvArgs gh = g;
                gh.w = find(hd, "wn.conv1D_" + ls + ".gate_f16")->ptr;
                if (planes_valid) {                                  // the res/skip layer in front left h as fp16 planes
                    gh.h_split = w.h16;
                    gh.h_split_ld = (C + 7) / 8 * 8;
                    gh.h_split_bstride = nsteps * (long long)gh.h_split_ld;
                }
                done = mbx::launch_wn_gate_f16(gh, stream);
                if (done) hd->last_gate_kernel[l] = MBX_GATE_K_SPLIT_F16;
                if (planes_only && !(done && planes_valid))
                    return fail(MBX_ERR_INVALID_ARGUMENT, "split precision: a gate layer did not take the plane-only hidden state");
            }
            bool use4 = !done && hd->winograd == 4 && !st_in && !st_out;
            if (use4 && d > 16) {
                // Dilations above 16 (the reference's default depth reaches 2048, custom_AE_layers.py:229-233): F(4,3) over the
                // d / 16 interleaved sub-sequences of every item (wn_winograd4w.hip, VS kernels).  A block covers 256 (128) rows
                // of ONE sub-sequence, so short items pad: cost in 256-row block units, the product-split block 0.56 of one
                // (half the products, ~12 % slower per product), the direct form twice the multiplies of the unpadded rows.
                // Both block shapes give the same bits; whether F(4,3) or the direct form runs depends on the launch only
                // under the default policy (batch_invariant: always F(4,3)).
                const int vs = d / 16;
                const long long vrows = ((long long)gs.max_rows + vs - 1) / vs, tiles = (C + 31) / 32;
                // (sub-sequences of at most 128 rows: a 256-row block takes two of them)
                const double cost_full = (vrows <= 128 ? 0.5 : (double)((vrows + 255) / 256)) * B * vs * tiles;
                const double cost_half = 0.56 * (double)((vrows + 127) / 128) * B * vs * tiles;
                const double cost_direct = 2.0 * ((double)gs.max_rows / 256.0) * B * tiles;
                if (hd->gate_small_shape >= 0 && !hd->winograd4_always) {
                    shape4 = hd->gate_small_shape;              // mbx_config.tune_gate_shape pins the shape (and F(4,3) itself)
                    split4 = shape4 != 0;
                } else {
                    split4 = split4 || cost_half < 0.95 * cost_full;
                    shape4 = split4 ? 1 : 0;
                    if (!hd->winograd4_always && std::min(cost_full, cost_half) > 0.9 * cost_direct) use4 = false;
                }
            }
            const DevTensor *wino4 = use4 ? find(hd, "wn.conv1D_" + ls + ".wino4w") : nullptr;
            if (wino4 && wino4->ndim == 3 && wino4->shape[0] == (C + 31) / 32 && wino4->shape[1] == (C + 7) / 8 &&
                wino4->shape[2] == 3072 && gs.cphase == 0) {
                mbx::ConvArgs gw = g;
                gw.w = wino4->ptr;
                // (the product-split shape holds 16 conditioning rows: cond_up >= 10; the 256-row shape takes cond_up >= 5)
                int ran = shape4;
                done = mbx::launch_wn_gate_winograd4w(gw, shape4, stream);
                if (!done && split4) {
                    ran = 0;
                    done = mbx::launch_wn_gate_winograd4w(gw, 0, stream);
                }
                if (done)
                    hd->last_gate_kernel[l] = d > 16 ? (ran ? MBX_GATE_K_F43_STRIDED_PSPLIT : MBX_GATE_K_F43_STRIDED)
                                                     : (ran == 2 ? MBX_GATE_K_F43_HSPLIT : ran == 1 ? MBX_GATE_K_F43_PSPLIT : MBX_GATE_K_F43);
            }
            // F(2,3): wave-tiled kernel on v_mfma_f32_16x16x4_f32 (wn_winograd2w.hip): streams, per-layer regions, MBX_CONV_F23
            const DevTensor *wino = (!done && hd->winograd) ? find(hd, "wn.conv1D_" + ls + ".wino2w") : nullptr;
            if (wino && wino->ndim == 3 && wino->shape[0] == (C + 31) / 32 && wino->shape[1] == (C + 7) / 8 &&
                wino->shape[2] == 2048) {
                mbx::ConvArgs gw = g;
                gw.w = wino->ptr;
                done = mbx::launch_wn_gate_winograd2w(gw, stream);
                if (done) hd->last_gate_kernel[l] = MBX_GATE_K_F23;
            }
            if (!done && (gs.cphase != 0 || gs.out_rows != 0))
                return fail(MBX_ERR_UNSUPPORTED, "per-layer regions need the Winograd F(2,3) gate kernel");
            if (!done) {
                mbx::launch_conv1d(g, mbx::EPI_GATE, stream);
                hd->last_gate_kernel[l] = MBX_GATE_K_DIRECT;
            }
        }
        const bool last = (l == L - 1);
        const Span &rs = res_sp[l];
        if (fold) {
            // skip path folded into the end convolution: layers 0..L-2 update h and add a W_skip W_end to the n_out-wide
            // output accumulator; the last layer's contribution is added by the tail kernel below
            if (!last) {
                // layer 0 with the start convolution folded in: rows [a0 | x'] (C + 16 channels) x [Wr ; Ws'], h starts from the bias
                const bool ext = l == 0 && fold_start;
                const int cin_l = ext ? C + 16 : C;
                const DevTensor *fw = find(hd, "wn.res_skip_" + ls + (ext ? ".fold_start" : ".fold")),
                                *fb = find(hd, "wn.res_skip_" + ls + ".fold_b");
                mbx::ConvArgs r = conv_args(w.a + rs.row0 * cin_l, nsteps * cin_l, cin_l, rs.nf, rs.rpf, rs.max_rows, B, fw, fb, 1,
                                            cin_l, C + n_out, 1, 0, MBX_PAD_ZERO, nullptr, 0, 0);
                r.h_init = ext;
                r.channels = C;
                r.zeros = hd->zeros;
                r.h = w.h + rs.row0 * C;
                r.skip = w.wn_out + rs.row0 * n_out;
                r.skip_ld = n_out;
                r.skip_bstride = nsteps * n_out;
                r.hs_bstride = nsteps * C;
                r.skip_init = (l == 0);
                ScopedEvents ev(hd, (hd->split_f16 && (!ext || find(hd, "wn.res_skip_0.fold_start_f16"))) ? PROF_RES_SKIP_F16 : PROF_RES_SKIP, stream);
                // large launches (>= two rounds of the 512 resident 128-row blocks): one block owns all columns of its rows.
                // Like the gate kernels' block shape this follows the launch size only under the default policy: a pinned
                // form (MBX_CONV_DIRECT, MBX_CONV_F23, batch_invariant, streams) pins the kernel, so results do not depend on the batch they ran in.
                bool done = false;
                // opt-in split half precision (mbx_config.wn_precision): every launch size, every layer whose split image the
                // host supplied (layer 0 with the folded start convolution too: its rows [a0 | x'] have the image *.fold_start_f16)
                const DevTensor *f16w = hd->split_f16 ? find(hd, "wn.res_skip_" + ls + (ext ? ".fold_start_f16" : ".fold_f16")) : nullptr;
                if (f16w && f16w->count == (long long)((cin_l + 31) / 32) * 12 * 1024) {
                    mbx::ConvArgs rh = r;
                    rh.w = f16w->ptr;
                    rh.gate_act = c.wn_gate_activation;
                    if (hd->split_f16_gate && !active_frames) {      // the next layer's gate reads the new hidden state as fp16 planes
                        rh.h_split = w.h16;
                        rh.h_split_ld = (C + 7) / 8 * 8;
                        rh.h_split_bstride = nsteps * (long long)rh.h_split_ld;
                        rh.h_planes_only = planes_only ? 1 : 0;
                    }
                    done = mbx::launch_wn_resskip_f16(rh, stream);
                    planes_valid = done && rh.h_split != nullptr;
                    if (planes_only && !planes_valid)
                        return fail(MBX_ERR_INVALID_ARGUMENT, "split precision: a res/skip layer did not take the plane-only hidden state");
                } else {
                    planes_valid = false;
                    if (planes_only) return fail(MBX_ERR_INVALID_ARGUMENT, "split precision: a res/skip image is missing behind a plane-only layer");
                }
                const DevTensor *fww = done ? nullptr : find(hd, "wn.res_skip_" + ls + (ext ? ".fold_start_wide" : ".fold_wide"));
                const long long wide_blocks = ((nsteps + 127) / 128) * B;
                const int npair = (C + n_out + 31) / 32;
                if (fww && fww->ndim == 3 && fww->shape[0] == (cin_l + 7) / 8 && fww->shape[1] == npair && fww->shape[2] == 256 &&
                    (hd->winograd4_always || (hd->winograd == 4 && !st_in && !st_out && wide_blocks >= 2 * 512))) {
                    mbx::ConvArgs rw = r;
                    rw.w = fww->ptr;
                    done = mbx::launch_wn_resskip_wide(rw, stream);
                }
                // small launches: wave-granular tiles (wn_resskip_wave.hip).  The form pinned by the streams and by
                // MBX_CONV_F23 (F(2,3) gate) runs this kernel at every size: an output's arithmetic does not depend on
                // its cut, so windows, per-layer regions and whole utterances agree bit for bit.
                const DevTensor *fwv = done ? nullptr : find(hd, "wn.res_skip_" + ls + (ext ? ".fold_start_wave" : ".fold_wave"));
                const long long wave_tiles = ((long long)rs.max_rows + 15) / 16 * B;
                const bool pinned23 = hd->winograd == 2 || (hd->winograd != 0 && (st_in || st_out));
                if (fwv && fwv->ndim == 3 && fwv->shape[0] == (cin_l + 15) / 16 && fwv->shape[1] == 12 && fwv->shape[2] == 512 &&
                    (pinned23 || (hd->winograd == 4 && !hd->winograd4_always && wave_tiles <= hd->resskip_wave_tiles))) {
                    mbx::ConvArgs rw = r;
                    rw.w = fwv->ptr;
                    rw.tune_split = hd->resskip_split;
                    done = mbx::launch_wn_resskip_wave(rw, stream);
                }
                if (!done && !mbx::launch_wn_resskip(r, stream)) return fail(MBX_ERR_INVALID_ARGUMENT, "folded res/skip layer does not fit its kernel");
            }
            continue;
        }
        mbx::ConvArgs r = conv_args(w.a + rs.row0 * C, nsteps * C, C, rs.nf, rs.rpf, rs.max_rows, B,
                                    find(hd, "wn.res_skip_" + ls + ".w"), find(hd, "wn.res_skip_" + ls + ".b"), 1, C,
                                    last ? C : 2 * C, 1, 0, MBX_PAD_ZERO, nullptr, 0, 0);
        r.channels = C;
        r.zeros = hd->zeros;
        r.h = w.h + rs.row0 * C;
        r.skip = w.skip + rs.row0 * C;
        r.hs_bstride = nsteps * C;
        r.skip_init = (l == 0);
        r.last_layer = last;
        {
            ScopedEvents ev(hd, PROF_RES_SKIP, stream);
            // LDS-DMA kernel with host-packed weights
            const DevTensor *pk = find(hd, "wn.res_skip_" + ls + ".packed");
            const int cout_l = last ? C : 2 * C;
            bool done = false;
            if (pk && pk->ndim == 3 && pk->shape[0] == (cout_l + 127) / 128 && pk->shape[1] == (C + 15) / 16 &&
                pk->shape[2] == 2048) {
                mbx::ConvArgs rp = r;
                rp.w = pk->ptr;
                done = mbx::launch_wn_resskip(rp, stream);
            }
            if (!done) mbx::launch_conv1d(r, mbx::EPI_RESSKIP, stream);
        }
    }
    {
        ScopedEvents ev(hd, PROF_TAIL, stream);
        const DevTensor *we = find(hd, "wn.end.w"), *be = find(hd, "wn.end.b"), *wpn = find(hd, "post.w"),
                        *bpn = find(hd, "post.b");
        const Span &ts = tail_sp;
        float *acc_t = w.wn_out + ts.row0 * n_out, *sub_t = w.sub + ts.row0 * M, *skip_t = w.skip + ts.row0 * C;
        if (fold) {
            const DevTensor *tw = find(hd, "wn.tail.fold"), *tb = find(hd, "wn.tail.fold_b");
            if (!mbx::launch_wn_tail(w.a + ts.row0 * C, nsteps * C, ts.nf, ts.rpf, ts.max_rows, B, C, tw->ptr, tb->ptr,
                                     n_out, wpn->ptr, bpn ? bpn->ptr : nullptr, M, L > 1 ? acc_t : nullptr,
                                     acc_t, nsteps * n_out, sub_t, nsteps * M, stream))
                return fail(MBX_ERR_INVALID_ARGUMENT, "folded WaveNet tail does not fit its kernel");
        }
        const DevTensor *wep = find(hd, "wn.end.packed");
        const bool fused = fold || (wep && wep->count == (long long)((C + 7) / 8) * 256 &&
            mbx::launch_wn_tail(skip_t, nsteps * C, ts.nf, ts.rpf, ts.max_rows, B, C, wep->ptr,
                                be ? be->ptr : nullptr, n_out, wpn->ptr, bpn ? bpn->ptr : nullptr, M, nullptr,
                                acc_t, nsteps * n_out, sub_t, nsteps * M, stream));
        if (!fused) {
            mbx::ConvArgs a = conv_args(skip_t, nsteps * C, C, ts.nf, ts.rpf, ts.max_rows, B, we, be, 1, C,
                                        n_out, 1, 0, MBX_PAD_ZERO, acc_t, nsteps * n_out, n_out);
            mbx::launch_conv1d(a, mbx::EPI_LINEAR, stream);
            // post-net 1x1 (reference custom_pulsed_generator.py:490-493,913-914)
            mbx::ConvArgs pn = conv_args(acc_t, nsteps * n_out, n_out, ts.nf, ts.rpf, ts.max_rows, B, wpn, bpn, 1, n_out, M, 1, 0,
                                         MBX_PAD_ZERO, sub_t, nsteps * M, M);
            mbx::launch_conv1d(pn, mbx::EPI_LINEAR, stream);
        }
    }
    return MBX_OK;
    };
    {
        const mbx_status wst = hd->blocks.empty() ? single_block() : run_wavenet_blocks(hd, w, B, T, n_frames, noise, stream);
        if (wst != MBX_OK) return wst;
    }
    // ---- sub-band gains instead of the STFT-domain filter (ps_use_stft: false; reference :857-884, 670, 916-917)
    if (c.ps_subband_gain) {
        if (active_frames || st_in || st_out) return fail(MBX_ERR_UNSUPPORTED, "ps_use_stft: false models run whole items only");
        auto lh = hd->lerp[c.hop_size];
        mbx::launch_subband_gain(w.sub, nsteps * M, w.ceps, (long long)T * c.n_ceps, n_frames, T, c.steps_per_frame, B, M, c.hop_size,
                                 lh.first, lh.second, c.spect_preserve_energy, stream);
    }
    // ---- sub-band rows carried between the ticks of a stream: rows in front of the WaveNet region come from the
    // caller's store (computed by the previous tick), the rows the next tick will need go there
    if (sub_carry) {
        mbx::launch_sub_carry(w_base.sub, nsteps * M, sub_store, (long long)sub_store_rows * M, sub_carry, B, sub_store_rows,
                              M, 0, stream);
        mbx::launch_sub_carry(w_base.sub, nsteps * M, sub_store, (long long)sub_store_rows * M, sub_carry, B, sub_store_rows,
                              M, 1, stream);
    }
    // the stages below see the active region (which contains the WaveNet's) as the item
    float *sub_act = w_base.sub + act0 * c.steps_per_frame * M;
    float *exc_act = w_base.exc + act0 * c.hop_size;
    float *ceps_act = w_base.ceps + act0 * c.n_ceps;
    float *f0_act = w_base.f0 + act0 * c.pulse_per_frame;
    int *cidx_act = w_base.ceps_index + act0;
    float *frames_act = w_base.frames + act0 * c.stft_win;
    float *audio_act = audio + act0 * c.hop_size;
    // ---- PQMF synthesis (reference :920-921)
    {
        ScopedEvents ev(hd, PROF_PQMF, stream);
        if (c.no_pqmf)   // pp_mod_subnet_use_pqmf: false -- source_signal = reshape(x, (B, rows * M)) (reference :922-923)
            mbx::launch_activation(sub_act, nsteps * M, n_frames_act, c.hop_size, act_frames_max * c.hop_size, B, 1, MBX_ACT_LINEAR,
                                   1.f, 0.f, exc_act, (long long)T * c.hop_size, stream);
        else
            mbx::launch_pqmf(sub_act, nsteps * M, n_frames_act, c.steps_per_frame, act_frames_max * c.steps_per_frame, B, M, hd->poly, hd->poly_t, hd->poly_ndm,
                             hd->poly_dm_min, exc_act, (long long)T * c.hop_size, stream);
    }
    if (c.ps_off || c.ps_subband_gain) {
        // ps_off / sub-band gains: signal = generate_excitation(...) (reference :663-672); samples behind an item's own length are zero
        ScopedEvents ev(hd, PROF_OVERLAP_ADD, stream);
        if (hipMemsetAsync(audio, 0, (size_t)B * T * c.hop_size * sizeof(float), stream) != hipSuccess)
            return fail(MBX_ERR_HIP, "hipMemsetAsync of the audio failed");
        mbx::launch_activation(exc_act, (long long)T * c.hop_size, n_frames_act, c.hop_size, act_frames_max * c.hop_size, B, 1,
                               MBX_ACT_LINEAR, 1.f, 0.f, audio_act, (long long)T * c.hop_size, stream);
    } else {
    // ---- STFT-domain filtering with the spectral envelope (reference :681-724, 801-855)
    // (the lifter row of a frame is selected from the F0 contour inside the kernel, reference :507-525)
    {
        ScopedEvents ev(hd, PROF_STFT_FILTER, stream);
        mbx::launch_stft_filter(sc, exc_act, (long long)T * c.hop_size, ceps_act, (long long)T * c.n_ceps, nullptr,
                                c.n_ceps_windows ? f0_act : nullptr, npulse, c.n_ceps_windows ? cidx_act : nullptr,
                                n_frames_act, T, B, frames_act, stream);
    }
    {
        ScopedEvents ev(hd, PROF_OVERLAP_ADD, stream);
        mbx::launch_overlap_add(sc, frames_act, n_frames_act, T, T - (int)act0, B, audio_act, (long long)T * c.hop_size, stream);
    }
    }
    if (nm_gain_src) {
        ScopedEvents ev(hd, PROF_NORM_MEL, stream);
        mbx::launch_norm_mel_gain(norm_mel_consts(hd), nm_gain_src, n_frames, T, B, audio, (long long)T * c.hop_size,
                                  false, stream);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MBX_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));

    auto &sg = hd->stages;
    if (nm_gain_src) sg["mel_norm"] = {w_base.mel_norm, (long long)T * c.mel_channels, (long long)T * c.mel_channels};
    else sg.erase("mel_norm");
    sg["f0"] = {w_base.f0, npulse, npulse};
    sg["pulse"] = {pulse_osc, npulse * (1 + c.wt_subharm_channels), npulse * (1 + c.wt_subharm_channels)};
    sg["cond"] = {w_base.cond, (long long)T * cond_cout, (long long)T * cond_cout};
    if (planes_only_run) {
        // split precision with the planes as the hidden state: the float32 tensor was not written; per row [C8 hi halves | C8 lo'
        // halves] = C8 float32 words (engine.py::stage rebuilds hi + 2^-11 lo' from it)
        sg.erase("wn_hidden");
        sg["wn_hidden_planes"] = {w_base.h16, nsteps * (long long)((C + 7) / 8 * 8), nsteps * (long long)((C + 7) / 8 * 8)};
    } else {
        sg.erase("wn_hidden_planes");
        sg["wn_hidden"] = {w_base.h, nsteps * C, nsteps * C};
    }
    if (!hd->fold_skip) sg["wn_skip"] = {w_base.skip, nsteps * C, nsteps * C};
    else sg.erase("wn_skip");
    sg["wn_out"] = {w_base.wn_out, nsteps * c.wn_out_channels, nsteps * c.wn_out_channels};
    sg["subbands"] = {w_base.sub, nsteps * M, nsteps * M};
    sg["excitation"] = {w_base.exc, (long long)T * c.hop_size, (long long)T * c.hop_size};
    sg["cepstrum"] = {w_base.ceps, (long long)T * c.n_ceps, (long long)T * c.n_ceps};
    sg["ceps_index"] = {w_base.ceps_index, (long long)T, (long long)T};
    sg["frames"] = {w_base.frames, (long long)T * c.stft_win, (long long)T * c.stft_win};
    return MBX_OK;
}

// ---- form of the dilated convolution -----------------------------------------------------------------------------------
// A Winograd form is available when the host supplied its weight images for every layer that runs a gate kernel, the
// padding is SAME and the kernel size 3 (layers whose dilation does not fit the kernels fall back per layer).
static bool form_available(const mbx_handle *hd, int form) {
    const mbx_config &c = hd->cfg;
    if (form == MBX_CONV_DIRECT) return true;
    if (form != MBX_CONV_F23 && form != MBX_CONV_F43) return false;
    if (c.wn_causal || c.wn_kernel_size != 3) return false;
    const bool f43 = form == MBX_CONV_F43;
    auto images = [&](const std::string &prefix, int C, int l0) {
        if (l0 >= c.wn_layers) return false;
        for (int l = l0; l < c.wn_layers; ++l) {
            const DevTensor *t = find(hd, prefix + "conv1D_" + std::to_string(l) + (f43 ? ".wino4w" : ".wino2w"));
            if (!t || t->ndim != 3 || t->shape[0] != (C + 31) / 32 || t->shape[1] != (C + 7) / 8 || t->shape[2] != (f43 ? 3072 : 2048))
                return false;
        }
        return true;
    };
    if (!hd->blocks.empty()) {
        if (!f43) return false;             // the block runner knows the F(4,3) and the direct form
        for (const auto &blk : hd->blocks)
            if (!images(blk.prefix, blk.C, 0)) return false;
        return true;
    }
    return images("wn.", c.wn_channels, hd->fold_start ? 1 : 0);
}

static void set_form(mbx_handle *hd, int form) {
    hd->winograd = form == MBX_CONV_F43 ? 4 : form == MBX_CONV_F23 ? 2 : 0;
    if (hd->cfg.wn_causal) hd->winograd = 0;  // causal padding: the direct form (generic kernel)
    hd->winograd4_always = hd->winograd == 4 && hd->cfg.batch_invariant != 0;
}

static int current_form(const mbx_handle *hd) {
    return hd->winograd == 4 ? MBX_CONV_F43 : hd->winograd == 2 ? MBX_CONV_F23 : MBX_CONV_DIRECT;
}

// One calibration: the same input through the direct form, F(4,3) and F(2,3); the fastest form whose audio differs from
// the direct form's by at most calib_fraction of the parity budget 1e-4 * max(1, |audio|) is adopted.  The difference
// between two float32 forms measures the rounding of the less exact one (F(4,3): ~5x the direct form's, growing with the
// amplitude of the residual stream).  Synchronises.
// forms: compare the convolution forms and adopt the fastest one within the threshold (MBX_CONV_AUTO at mbx_create,
// mbx_calibrate); false: only the split-precision check of a handle whose form the configuration pins
static mbx_status calibrate_run(mbx_handle *hd, const float *mel, const int32_t *n_frames, int B, int T, const float *noise,
                                void *workspace, size_t workspace_bytes, hipStream_t stream, int kind, bool forms = true) {
    const size_t n = (size_t)B * T * hd->cfg.hop_size;
    float *audio_dev = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&audio_dev), n * sizeof(float)));
    std::vector<float> ref(n), got(n);
    const bool was_profiling = hd->profiling;
    hd->profiling = false;
    const int form_before = current_form(hd);
    // the opt-in split precision is measured like a form: every reference and form run below is float32; the handle's
    // configuration in split precision is then held to the same threshold against the float32 direct form
    const bool split_req = hd->split_f16, split_gate_req = hd->split_f16_gate;
    hd->split_f16 = hd->split_f16_gate = false;
    auto run = [&](int form, std::vector<float> &host) -> mbx_status {
        set_form(hd, form);
        mbx_status st = forward_impl(hd, mel, n_frames, B, T, noise, audio_dev, workspace, workspace_bytes, stream);
        if (st != MBX_OK) return st;
        if (hipMemcpyAsync(host.data(), audio_dev, n * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess)
            return fail(MBX_ERR_HIP, "calibration: reading the audio back failed");
        return MBX_OK;
    };
    auto done = [&](mbx_status st) {
        (void)hipFree(audio_dev);
        hd->profiling = was_profiling;
        // the calibration forwards filled the stage table with pointers into their own (temporary, or by now overwritten)
        // workspace and strides of the calibration batch: mbx_stage answers "unknown stage" until the caller's next forward
        hd->stages.clear();
        hd->last_gate_layers = 0;
        if (st != MBX_OK) {
            set_form(hd, form_before);
            hd->split_f16 = split_req;
            hd->split_f16_gate = split_gate_req;
        }
        return st;
    };
    mbx_status st = run(MBX_CONV_DIRECT, ref);
    if (st != MBX_OK) return done(st);
    float ref_max = 0.f;
    bool finite = true;
    for (float v : ref) {
        if (!std::isfinite(v)) finite = false;
        ref_max = std::max(ref_max, std::fabs(v));
    }
    const float frac = hd->cfg.calib_fraction > 0.f ? hd->cfg.calib_fraction : 0.25f;
    const float threshold = frac * 1e-4f * std::max(1.f, ref_max);
    float err[2] = {-1.f, -1.f};
    const int form_list[2] = {MBX_CONV_F43, MBX_CONV_F23};
    for (int k = 0; k < 2 && forms; ++k) {
        if (!form_available(hd, form_list[k])) continue;
        st = run(form_list[k], got);
        if (st != MBX_OK) return done(st);
        float e = 0.f;
        for (size_t i = 0; i < n; ++i) {
            const float d = std::fabs(got[i] - ref[i]);
            e = std::isfinite(d) ? std::max(e, d) : INFINITY;
        }
        err[k] = e;
    }
    int form = forms ? MBX_CONV_DIRECT : form_before;
    if (forms && finite && err[0] >= 0.f && err[0] <= threshold) form = MBX_CONV_F43;
    else if (forms && finite && err[1] >= 0.f && err[1] <= threshold) form = MBX_CONV_F23;
    set_form(hd, form);
    if (forms) {
        hd->calibrated = kind;
        hd->calib_err43 = err[0];
        hd->calib_err23 = err[1];
    }
    hd->calib_ref = ref_max;
    hd->calib_threshold = threshold;
    if (split_req) {
        // the handle as it will run (its form, split precision on) against the float32 direct form
        hd->split_f16 = true;
        hd->split_f16_gate = split_gate_req;
        st = run(form, got);
        if (st != MBX_OK) return done(st);
        float e = 0.f;
        for (size_t i = 0; i < n; ++i) {
            const float d = std::fabs(got[i] - ref[i]);
            e = std::isfinite(d) ? std::max(e, d) : INFINITY;
        }
        hd->calib_err_split = e;
        // (the split precision rides on the form's own error: its budget is what the threshold leaves)
        hd->split_rejected = !(finite && e <= threshold);
        if (hd->split_rejected) hd->split_f16 = hd->split_f16_gate = false;
    }
    return done(MBX_OK);
}

// MBX_CONV_AUTO at mbx_create: two items of 40 frames of seeded synthetic log-mel input -- one with independent values
// of the level statistics the models are fed with (N(-5, 2^2) log amplitudes, clipped like scale_mel's output), one a
// smooth loud sweep that drives the conditioning towards the saturated side of the gates -- and a seeded noise draw.
// What is measured is this handle's own weights on plausible input, not the user's data: mbx_calibrate does that.
static mbx_status calibrate_on_synthetic_mel(mbx_handle *hd, bool forms) {
    const mbx_config &c = hd->cfg;
    const int B = 2, T = 40;
    uint64_t rs = 0x9E3779B97F4A7C15ull;
    auto uni = [&]() {                       // xorshift64*: (0, 1]
        rs ^= rs >> 12;
        rs ^= rs << 25;
        rs ^= rs >> 27;
        return (double)(((rs * 0x2545F4914F6CDD1Dull) >> 11) + 1) / 9007199254740992.0;
    };
    auto gauss = [&]() { return std::sqrt(-2.0 * std::log(uni())) * std::cos(2.0 * M_PI * uni()); };
    std::vector<float> mel((size_t)B * T * c.mel_channels), noise((size_t)B * T * c.steps_per_frame);
    for (int t = 0; t < T; ++t)
        for (int m = 0; m < c.mel_channels; ++m) {
            const double v0 = std::log(std::exp(-5.0 + 2.0 * gauss()) + 1e-5);
            const double v1 = -3.0 + 4.0 * std::sin(0.21 * t + 0.08 * m) + 1.5 * std::cos(0.045 * m * (1 + t % 7)) + 0.3 * gauss();
            mel[((size_t)0 * T + t) * c.mel_channels + m] = (float)std::min(2.0, std::max(-11.5, v0));
            mel[((size_t)1 * T + t) * c.mel_channels + m] = (float)std::min(2.0, std::max(-11.5, v1));
        }
    for (float &v : noise) v = (float)gauss();
    const size_t ws_bytes = mbx_workspace_size(hd, B, T);
    float *mel_dev = nullptr, *noise_dev = nullptr;
    void *ws = nullptr;
    auto release = [&](mbx_status st) {
        if (mel_dev) (void)hipFree(mel_dev);
        if (noise_dev) (void)hipFree(noise_dev);
        if (ws) (void)hipFree(ws);
        return st;
    };
    if (hipMalloc(reinterpret_cast<void **>(&mel_dev), mel.size() * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&noise_dev), noise.size() * sizeof(float)) != hipSuccess ||
        hipMalloc(&ws, ws_bytes) != hipSuccess ||
        hipMemcpy(mel_dev, mel.data(), mel.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(noise_dev, noise.data(), noise.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return release(fail(MBX_ERR_HIP, "calibration: device allocation / upload failed"));
    return release(calibrate_run(hd, mel_dev, nullptr, B, T, c.noise_sigma != 0.f ? noise_dev : nullptr, ws, ws_bytes, nullptr, 1, forms));
}

mbx_status mbx_calibrate(mbx_handle *hd, const float *mel, const int32_t *n_frames, int32_t batch, int32_t max_frames,
                         const float *noise, void *workspace, size_t workspace_bytes, void *hip_stream) {
    if (!hd || !mel || !workspace) return fail(MBX_ERR_INVALID_ARGUMENT, "null argument");
    if (batch <= 0 || max_frames <= 0) return fail(MBX_ERR_INVALID_ARGUMENT, "batch and max_frames must be positive");
    DeviceGuard guard(hd->device);
    if (!guard.ok) return fail(MBX_ERR_HIP, "cannot select the handle's device");
    return calibrate_run(hd, mel, n_frames, batch, max_frames, noise, workspace, workspace_bytes,
                         static_cast<hipStream_t>(hip_stream), 2);
}

mbx_status mbx_conv_form(const mbx_handle *hd, mbx_conv_form_info *info) {
    if (!hd || !info || info->struct_size != (int32_t)sizeof(mbx_conv_form_info))
        return fail(MBX_ERR_INVALID_ARGUMENT, "mbx_conv_form_info ABI mismatch (struct_size)");
    info->requested = hd->cfg.wn_conv_form;
    info->form = current_form(hd);
    info->stream_form = hd->winograd != 0 && form_available(hd, MBX_CONV_F23) ? MBX_CONV_F23 : MBX_CONV_DIRECT;
    info->calibrated = hd->calibrated;
    info->batch_invariant = (hd->winograd != 4 || hd->winograd4_always) ? 1 : 0;
    info->fold_skip = hd->fold_skip;
    info->fold_start = hd->fold_start;
    info->split_f16_layers = 0;
    for (int l = 0; hd->split_f16 && l + 1 < hd->cfg.wn_layers; ++l)
        info->split_f16_layers += find(hd, "wn.res_skip_" + std::to_string(l) + (l == 0 && hd->fold_start ? ".fold_start_f16" : ".fold_f16")) != nullptr;
    info->split_f16_gate_layers = hd->split_f16_gate ? std::max(0, hd->cfg.wn_layers - 1) : 0;
    info->err_split = hd->calib_err_split;
    info->split_rejected = hd->split_rejected;
    info->err_f43 = hd->calib_err43;
    info->err_f23 = hd->calib_err23;
    info->ref_max = hd->calib_ref;
    info->threshold = hd->calib_threshold;
    info->f0_float64_chain = hd->f0_full64 ? 1 : 0;
    info->n_gate_layers = hd->last_gate_layers;
    for (int l = 0; l < MBX_MAX_WN_LAYERS; ++l) info->gate_kernel[l] = l < hd->last_gate_layers ? hd->last_gate_kernel[l] : MBX_GATE_K_NONE;
    return MBX_OK;
}

mbx_status mbx_forward(mbx_handle *hd, const float *mel, const int32_t *n_frames, int32_t batch, int32_t max_frames,
                       const float *noise, float *audio, void *workspace, size_t workspace_bytes, void *hip_stream) {
    return forward_impl(hd, mel, n_frames, batch, max_frames, noise, audio, workspace, workspace_bytes, hip_stream);
}

mbx_status mbx_forward_stream(mbx_handle *hd, const float *mel, const int32_t *n_frames, int32_t batch,
                              int32_t max_frames, const float *noise, float *audio, void *workspace,
                              size_t workspace_bytes, const mbx_stream_state *state_in, mbx_stream_state *state_out,
                              void *hip_stream) {
    static_assert(sizeof(mbx_stream_state) == sizeof(mbx::StreamState), "stream state layout");
    if (!state_in) return fail(MBX_ERR_INVALID_ARGUMENT, "state_in is required (use mbx_forward for whole utterances)");
    ForwardExtras ex;
    ex.st_in = reinterpret_cast<const mbx::StreamState *>(state_in);
    ex.st_out = reinterpret_cast<mbx::StreamState *>(state_out);
    return forward_impl(hd, mel, n_frames, batch, max_frames, noise, audio, workspace, workspace_bytes, hip_stream, ex);
}

mbx_status mbx_forward_ex(mbx_handle *hd, const float *mel, const int32_t *n_frames, int32_t batch, int32_t max_frames,
                          const float *noise, float *audio, void *workspace, size_t workspace_bytes,
                          const mbx_forward_options *options, void *hip_stream) {
    if (!options || options->struct_size != (int32_t)sizeof(mbx_forward_options))
        return fail(MBX_ERR_INVALID_ARGUMENT, "mbx_forward_options ABI mismatch (struct_size)");
    if (!(options->transposition > 0.f)) return fail(MBX_ERR_INVALID_ARGUMENT, "transposition must be positive");
    if ((!options->layer_carry && options->layer_rows != 0) || options->layer_rows < 0)
        return fail(MBX_ERR_INVALID_ARGUMENT, "layer_rows must be >= 0 and needs layer_carry");
    const LayerOpts lay{options->layer_store, options->layer_store_floats, options->layer_carry, options->layer_rows};
    ForwardExtras ex;
    ex.st_in = reinterpret_cast<const mbx::StreamState *>(options->state_in);
    ex.st_out = reinterpret_cast<mbx::StreamState *>(options->state_out);
    ex.f0_in = options->f0;
    ex.transposition = options->transposition;
    ex.active_begin = options->active_begin;
    ex.active_frames = options->active_frames;
    ex.wn_begin = options->wn_begin;
    ex.wn_frames = options->wn_frames;
    ex.sub_store = options->sub_store;
    ex.sub_store_rows = options->sub_store_rows;
    ex.sub_carry = options->sub_carry;
    ex.active_max_frames = options->active_max_frames;
    ex.wn_max_frames = options->wn_max_frames;
    ex.lay = options->layer_carry ? &lay : nullptr;
    ex.fe_store = options->fe_store;
    ex.fe_ring_frames = options->fe_ring_frames;
    ex.fe_pos = options->fe_pos;
    ex.fe_new_frames = options->fe_new_frames;
    ex.fe_margin_frames = options->fe_margin_frames;
    ex.fe_end_frames = options->fe_end_frames;
    return forward_impl(hd, mel, n_frames, batch, max_frames, noise, audio, workspace, workspace_bytes, hip_stream, ex);
}

mbx_status mbx_window_advance(mbx_handle *hd, float *mel_window, const float *mel_new, float *noise_window,
                              const float *noise_new, int32_t batch, int32_t frames, int32_t step_frames, void *hip_stream) {
    if (!hd || !mel_window || !mel_new || (noise_window != nullptr) != (noise_new != nullptr))
        return fail(MBX_ERR_INVALID_ARGUMENT, "null argument (noise_window and noise_new go together)");
    DeviceGuard guard(hd->device);
    if (!guard.ok) return fail(MBX_ERR_HIP, "cannot select the handle's device");
    if (!mbx::launch_window_advance(mel_window, mel_new, noise_window, noise_new, batch, frames, step_frames,
                                    hd->cfg.mel_channels, hd->cfg.steps_per_frame, static_cast<hipStream_t>(hip_stream)))
        return fail(MBX_ERR_INVALID_ARGUMENT, "window advance: need 0 < step_frames <= frames and a window of at most 64 KB per item");
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_clock_probe(mbx_handle *hd, uint64_t *device_out4, int64_t real_ticks, void *hip_stream) {
    if (!hd || !device_out4 || real_ticks <= 0 || real_ticks > 100000000LL)
        return fail(MBX_ERR_INVALID_ARGUMENT, "clock probe: need a device buffer of 4 x uint64 and 0 < real_ticks <= 1e8 (1 s)");
    DeviceGuard guard(hd->device);
    if (!guard.ok) return fail(MBX_ERR_HIP, "cannot select the handle's device");
    mbx::launch_clock_probe(reinterpret_cast<unsigned long long *>(device_out4), (unsigned long long)real_ticks,
                            static_cast<hipStream_t>(hip_stream));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_window_update(mbx_handle *hd, float *mel_window, const float *mel_new, float *noise_window,
                             const float *noise_new, int32_t batch, int32_t window_frames, int32_t shift_frames,
                             int32_t keep_frames, int32_t new_frames, void *hip_stream) {
    if (!hd || !mel_window || !mel_new || (noise_window != nullptr) != (noise_new != nullptr))
        return fail(MBX_ERR_INVALID_ARGUMENT, "null argument (noise_window and noise_new go together)");
    DeviceGuard guard(hd->device);
    if (!guard.ok) return fail(MBX_ERR_HIP, "cannot select the handle's device");
    if (!mbx::launch_window_update(mel_window, mel_new, noise_window, noise_new, batch, window_frames, shift_frames, keep_frames,
                                   new_frames, hd->cfg.mel_channels, hd->cfg.steps_per_frame, static_cast<hipStream_t>(hip_stream)))
        return fail(MBX_ERR_INVALID_ARGUMENT, "window update: need shift + keep <= window, keep + new <= window and at most 64 KB kept per item");
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_emit_rows(mbx_handle *hd, const float *audio, int64_t row_floats, int32_t batch, int64_t first, int64_t count,
                         float *host_out, void *hip_stream) {
    if (!hd || !audio || !host_out || batch <= 0 || count <= 0 || first < 0 || first + count > row_floats)
        return fail(MBX_ERR_INVALID_ARGUMENT, "emit rows: need 0 <= first, first + count <= row_floats");
    DeviceGuard guard(hd->device);
    if (!guard.ok) return fail(MBX_ERR_HIP, "cannot select the handle's device");
    HIP_TRY(hipMemcpy2DAsync(host_out, (size_t)count * sizeof(float), audio + first, (size_t)row_floats * sizeof(float),
                             (size_t)count * sizeof(float), (size_t)batch, hipMemcpyDeviceToHost, static_cast<hipStream_t>(hip_stream)));
    return MBX_OK;
}

mbx_status mbx_mel_analysis(const float *audio, const int32_t *n_samples, int32_t batch, int32_t max_samples,
                            int32_t win, int32_t hop, int32_t fft_size, int32_t n_mels, const float *window,
                            const float *twiddle, const float *basis, const int32_t *bin_lo, const int32_t *bin_hi,
                            float eps, float *out, int32_t max_frames, void *hip_stream) {
    mbx::MelAnalysisArgs a{};
    a.audio = audio;
    a.audio_bstride = max_samples;
    a.n_samples = n_samples;
    a.max_samples = max_samples;
    a.batch = batch;
    a.win = win;
    a.hop = hop;
    a.fft_size = fft_size;
    a.n_mels = n_mels;
    a.window = window;
    a.twiddle = twiddle;
    a.basis = basis;
    a.bin_lo = bin_lo;
    a.bin_hi = bin_hi;
    a.eps = eps;
    a.out = out;
    a.max_frames = max_frames;
    if (!mbx::launch_mel_analysis(a, static_cast<hipStream_t>(hip_stream)))
        return fail(MBX_ERR_INVALID_ARGUMENT, "mel analysis: sizes do not fit the kernel (fft_size a power of two <= 2048, "
                                              "win <= fft_size, max_samples > win/2, max_frames >= max_samples/hop + 1)");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MBX_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return MBX_OK;
}

mbx_status mbx_profile_enable(mbx_handle *handle, int32_t enabled) {
    if (!handle) return fail(MBX_ERR_INVALID_ARGUMENT, "null argument");
    handle->profiling = enabled != 0;
    return MBX_OK;
}

mbx_status mbx_profile_read(mbx_handle *handle, const char *kernel, double *total_ms, int64_t *launches) {
    if (!handle || !kernel || !total_ms || !launches) return fail(MBX_ERR_INVALID_ARGUMENT, "null argument");
    int kind = -1;
    for (int k = 0; k < PROF_KINDS; ++k)
        if (std::strcmp(kernel, kProfNames[k]) == 0) kind = k;
    if (kind < 0) return fail(MBX_ERR_INVALID_ARGUMENT, std::string("unknown profile stage ") + kernel);
    double sum = 0.0;
    for (size_t i = 0; i < handle->ev_used[kind]; ++i) {
        auto &pr = handle->ev_pool[kind][i];
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)handle->ev_used[kind];
    handle->ev_used[kind] = 0;
    return MBX_OK;
}

mbx_status mbx_profile_read_launches(mbx_handle *handle, const char *kernel, float *launch_ms, int64_t capacity,
                                     int64_t *launches) {
    if (!handle || !kernel || !launches || (capacity > 0 && !launch_ms) || capacity < 0)
        return fail(MBX_ERR_INVALID_ARGUMENT, "null argument");
    int kind = -1;
    for (int k = 0; k < PROF_KINDS; ++k)
        if (std::strcmp(kernel, kProfNames[k]) == 0) kind = k;
    if (kind < 0) return fail(MBX_ERR_INVALID_ARGUMENT, std::string("unknown profile stage ") + kernel);
    for (size_t i = 0; i < handle->ev_used[kind]; ++i) {
        auto &pr = handle->ev_pool[kind][i];
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        if ((int64_t)i < capacity) launch_ms[i] = ms;
    }
    *launches = (int64_t)handle->ev_used[kind];
    handle->ev_used[kind] = 0;
    return MBX_OK;
}

mbx_status mbx_stage(const mbx_handle *handle, const char *name, const void **device_ptr, int64_t *count,
                     int64_t *stride) {
    if (!handle || !name || !device_ptr || !count || !stride) return fail(MBX_ERR_INVALID_ARGUMENT, "null argument");
    auto it = handle->stages.find(name);
    if (it == handle->stages.end()) return fail(MBX_ERR_INVALID_ARGUMENT, std::string("unknown stage ") + name);
    *device_ptr = it->second.ptr;
    *count = it->second.count;
    *stride = it->second.stride;
    return MBX_OK;
}

mbx_status mbx_pqmf_synthesis(mbx_handle *hd, const float *x, int32_t batch, int32_t n_steps, float *y,
                              void *hip_stream) {
    if (!hd || !x || !y || batch <= 0 || n_steps <= 0) return fail(MBX_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(hd->device);
    const int M = hd->cfg.subbands;
    mbx::launch_pqmf(x, (long long)n_steps * M, nullptr, 1, n_steps, batch, M, hd->poly, hd->poly_t, hd->poly_ndm,
                     hd->poly_dm_min, y, (long long)n_steps * M, static_cast<hipStream_t>(hip_stream));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_conv1d(mbx_handle *hd, const float *x, int32_t batch, int32_t n_rows, int32_t cin, const float *w,
                      const float *b, const float *alpha, int32_t ks, int32_t cout, int32_t dilation, int32_t pad_l,
                      int32_t pad_mode, float *y, void *hip_stream) {
    if (!hd || !x || !w || !y || batch <= 0 || n_rows <= 0 || cin <= 0 || cout <= 0 || ks <= 0 || dilation <= 0)
        return fail(MBX_ERR_INVALID_ARGUMENT, "bad argument");
    if (pad_mode < MBX_PAD_ZERO || pad_mode > MBX_PAD_EDGE) return fail(MBX_ERR_INVALID_ARGUMENT, "bad pad_mode");
    DeviceGuard guard(hd->device);
    DevTensor wt, bt;
    wt.ptr = const_cast<float *>(w);
    bt.ptr = const_cast<float *>(b);
    mbx::ConvArgs a = conv_args(x, (long long)n_rows * cin, cin, nullptr, 1, n_rows, batch, &wt, b ? &bt : nullptr, ks,
                                cin, cout, dilation, pad_l, pad_mode, y, (long long)n_rows * cout, cout);
    a.alpha = alpha;
    a.zeros = hd->zeros;
    // the way the launch sequence issues its mel-rate convolutions: large launches take the LDS-staged tile kernel
    mbx::launch_conv1d_group(&a, 1, static_cast<hipStream_t>(hip_stream));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_conv1d_f64acc(mbx_handle *hd, const float *x, int32_t batch, int32_t n_rows, int32_t cin, const float *w,
                             const float *b, const float *alpha, int32_t ks, int32_t cout, int32_t dilation, int32_t pad_l,
                             int32_t pad_mode, float *y, void *hip_stream) {
    if (!hd || !x || !w || !y || batch <= 0 || n_rows <= 0 || cin <= 0 || cout <= 0 || ks <= 0 || dilation <= 0)
        return fail(MBX_ERR_INVALID_ARGUMENT, "bad argument");
    if (pad_mode < MBX_PAD_ZERO || pad_mode > MBX_PAD_EDGE) return fail(MBX_ERR_INVALID_ARGUMENT, "bad pad_mode");
    if (cin % 4 || (reinterpret_cast<uintptr_t>(x) & 15))
        return fail(MBX_ERR_UNSUPPORTED, "the float64-accumulating convolution needs cin % 4 == 0 and a 16-byte aligned input");
    DeviceGuard guard(hd->device);
    DevTensor wt, bt;
    wt.ptr = const_cast<float *>(w);
    bt.ptr = const_cast<float *>(b);
    mbx::ConvArgs a = conv_args(x, (long long)n_rows * cin, cin, nullptr, 1, n_rows, batch, &wt, b ? &bt : nullptr, ks,
                                cin, cout, dilation, pad_l, pad_mode, y, (long long)n_rows * cout, cout);
    a.alpha = alpha;
    a.precise = 1;
    mbx::launch_conv1d_group(&a, 1, static_cast<hipStream_t>(hip_stream));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_lin_interp(mbx_handle *hd, const float *x, int32_t batch, int32_t n_rows, int32_t channels, int32_t up,
                          float *y, void *hip_stream) {
    if (!hd || !x || !y || batch <= 0 || n_rows <= 0 || channels <= 0) return fail(MBX_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(hd->device);
    auto it = hd->lerp.find(up);
    if (it == hd->lerp.end()) return fail(MBX_ERR_INVALID_ARGUMENT, "interpolation factor not part of this model");
    mbx::launch_lin_interp(x, (long long)n_rows * channels, nullptr, 1, n_rows, batch, channels, up, it->second.first,
                           it->second.second, MBX_ACT_LINEAR, 1.f, 0.f, y, (long long)n_rows * up * channels,
                           static_cast<hipStream_t>(hip_stream));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_wavetable(mbx_handle *hd, const float *f0, int32_t batch, int32_t n, float *pulse, float *phase,
                         float *scratch, void *hip_stream) {
    if (!hd || !f0 || !pulse || !scratch || batch <= 0 || n <= 0) return fail(MBX_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(hd->device);
    float *cum = scratch;
    float *chunk_last = scratch + (size_t)batch * n;
    mbx::launch_wavetable(wavetable_consts(hd), f0, n, nullptr, 1, n, batch, pulse, phase, cum, chunk_last, nullptr,
                          nullptr, static_cast<hipStream_t>(hip_stream));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_norm_mel(mbx_handle *hd, const float *mel, const int32_t *n_frames, int32_t batch, int32_t frames,
                        float *mel_out, float *gain, float *scratch, void *hip_stream) {
    if (!hd || !mel || !mel_out || !scratch || batch <= 0 || frames <= 0)
        return fail(MBX_ERR_INVALID_ARGUMENT, "bad argument");
    if (hd->cfg.nm_iters <= 0) return fail(MBX_ERR_INVALID_ARGUMENT, "this model has no RMS normalisation (nm_iters == 0)");
    DeviceGuard guard(hd->device);
    const mbx_config &c = hd->cfg;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    mbx::NormMelConsts k = norm_mel_consts(hd);
    const float *src = mbx::launch_norm_mel(k, mel, (long long)frames * c.mel_channels, n_frames, frames, batch, scratch,
                                            scratch + (size_t)batch * frames, mel_out, stream);
    if (gain)
        mbx::launch_norm_mel_gain(k, src, n_frames, frames, batch, gain, (long long)frames * c.hop_size, true, stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

mbx_status mbx_stft_filter(mbx_handle *hd, const float *excitation, const float *cepstrum, const int32_t *ceps_index,
                           int32_t batch, int32_t frames, float *audio, float *scratch, void *hip_stream) {
    if (!hd || !excitation || !cepstrum || !audio || !scratch || batch <= 0 || frames <= 0)
        return fail(MBX_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(hd->device);
    const mbx_config &c = hd->cfg;
    mbx::StftConsts sc = stft_consts(hd);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    mbx::launch_stft_filter(sc, excitation, (long long)frames * c.hop_size, cepstrum, (long long)frames * c.n_ceps,
                            c.n_ceps_windows ? ceps_index : nullptr, nullptr, 0, nullptr, nullptr, frames, batch, scratch,
                            stream);
    mbx::launch_overlap_add(sc, scratch, nullptr, frames, frames, batch, audio, (long long)frames * c.hop_size, stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MBX_OK : fail(MBX_ERR_HIP, hipGetErrorString(e));
}

}  // extern "C"
