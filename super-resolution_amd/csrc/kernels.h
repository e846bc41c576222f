// Shared declarations between the HIP kernels and the C-ABI host layer.
// gfx950 only; see DESIGN.md for the data layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sr {

enum Prec { P_FP32 = 0, P_BF16 = 1, P_BF16X3 = 2 };
enum Act { A_NONE = 0, A_RELU = 1, A_LRELU = 2, A_TANH = 3, A_CLIP01 = 4 };
enum ConvMode { M_CONV = 0, M_CONVT = 1 };

// Geometry of the conv kernels: a workgroup (256 threads = 4 waves) computes a
// TH x 32 pixel tile of NT*32 output channels; the (TH+2) x 34 halo tile of one
// input-channel chunk is staged in LDS, pixel-major, 16 B of padding per pixel.
constexpr int TW = 32;

struct ConvArgs {
    const void* in;        // [B,H,W,cin_p] activations (float or bf16)
    const void* wpack;     // packed weights, MFMA fragment order (see pack_weights)
    const float* bias;     // [ncb * NT*32] fp32, zero padded
    void* out;             // [B,Ho,Wo,out_cs]
    const void* res;       // residual added after the activation, or nullptr
    int B, H, W;           // input spatial dims (output = same, or 2x for M_CONVT)
    int cin_p;             // input pixel stride in elements (multiple of 16)
    int out_cs;            // output pixel stride in elements
    int res_cs;            // residual pixel stride in elements
    int cout_store;        // number of output channels actually stored
    int out_f32;           // 1: out/res are float, 0: bf16
    int act;
    int tiles_x, tiles_y;
    int nchunk;            // number of input-channel chunks
    int wstride_cb;        // 16-byte vectors of wpack per output-channel block
    // planar activations: [B][planes][H][W][64 bytes] instead of [B][H][W][channels]; a plane holds
    // one conv chunk (16 fp32 / 32 bf16 channels), so a chunk's halo row is contiguous full 128-B
    // lines instead of a 64-byte half of every pixel's lines.  Image stride is the same either way.
    int in_planar;         // input is planar (needs 64-byte chunks: steps == prec_maxsteps)
    int in_split;          // transposed conv, bf16x3: the planar input is split-planar (conv_trunk.hip): units of 8 bf16 (hi | lo)
                           // in permuted channel order go to the hi / lo LDS tiles as they are; weights packed with kperm
    int out_planar;        // output and residual are planar (cout_store a multiple of 64 bytes)
    // transposed conv only: packed 64 x 27 tap-projection weights of the following <= 3-channel conv
    // (conv_mfma.hip epilogue_proj); out is then the fp32 P tensor [B,2H,2W,32]; nullptr = plain epilogue
    const void* proj_w;
    int dbg;               // ablation switches for tools/conv_bench.py (0 in production)
    unsigned long long* stamps;   // diagnostic build only (dbg & 64): per-phase cycle sums, else nullptr
    // training data-gradient: res is not a residual but the stored output y of the activation `act`
    // that produced this conv's gradient target; out = conv(in) * act'(y) (1 where y > 0, else 0 / 0.2),
    // no bias activation of its own.  M_CONV, fp32 / bf16x3, vector-aligned strides only.
    int res_mask;
    // M_CONV on the LDS-transpose path: instead of the activated output, store its 2x2 / stride-2 max pooling
    // [B,H/2,W/2,out_cs] here (H, W even; `out` is not written).  FNet's encoder: Conv2D + LeakyReLU + MaxPooling2D,
    // reference layers.py:31-36 -- the full-resolution tensor has no other reader.
    void* pool_out;
    // transposed conv with proj_w: add the nine shifted taps in the kernel as well (conv_mfma.hip, end of convT3x3_lds_kernel).
    // gather_out [B,2H,2W,gather_cs] receives act(bias + sum) of every pixel whose 3x3 neighbourhood lies inside the workgroup's
    // 8 x 64 output tile; for the pixels on a tile's border, `out` takes four partial-sum images S[k][B][2H][2W][4 floats]
    // (k = 2 * (from the tile above/below) + (from the tile left/right)) that launch_gather_fix adds up.  nullptr: out = P
    float* gather_out;
    const float* gather_bias;   // [3], the gathered conv's bias
    int gather_cs, gather_cout, gather_act;   // pixel stride / stored channels (<= 3) / A_NONE, A_RELU or A_CLIP01
};

// steps per chunk / chunk geometry, shared by the packer and the kernel
__host__ __device__ constexpr int prec_ks(int prec) { return prec == P_FP32 ? 8 : 16; }      // channels per k-step
// conv: chunks of 16 (fp32, bf16x3) or 32 (bf16) input channels, so that the halo tile plus the
// chunk's weights (36 KiB) let two workgroups share a CU's LDS;
// transposed conv: all 64 input channels of this network's layers in ONE chunk
__host__ __device__ constexpr int prec_maxsteps(int prec, int mode) {
    return mode == M_CONV ? (prec == P_BF16X3 ? 1 : 2) : (prec == P_FP32 ? 8 : 4);
}

// taps-in-N kernel for <= 3 output channels (conv_tapsn.hip): channels per chunk = one 128-B line per pixel
__host__ __device__ constexpr int tapsn_chunk(int prec) { return prec == P_BF16 ? 64 : 32; }

// SRNet trunk conv on split-planar activations (conv_trunk.hip): 64 -> 64 channels, bf16x3 arithmetic.
// Tensors are [B][4 planes][H][W][64 B]; see conv_trunk.hip for the record layout.
struct TrunkArgs {
    const unsigned char* in;
    unsigned char* out;
    const unsigned char* res;     // residual added after the activation (may be `out`), or nullptr
    const unsigned char* wpack;   // [2 halves][4 chunks][9 taps][hi, lo][64 lanes] x 16 B (sr_api.hip trunk_entries)
    const float* bias;            // [64]
    unsigned zero_off;            // byte offset from `in` of 64 zero bytes (behind the tensor): the source of halo pixels outside the image
    unsigned out_tail;            // launch_trunk_convT only: byte offset from `out` of the tail behind the (larger) output tensor
    int B, H, W;
    int relu;                     // ReLU after the bias (before the residual)
    int tiles_x, tiles_y;         // set by the launcher
    int dbg;                      // ablation switches for the micro-benchmark (0 in production): 1 no tile DMA, 2 no MFMA loop, 4 no epilogue memory traffic
    int out_mx6 = 0;              // launch_trunk_conv, ReLU layer: `out` takes the mx6 format of conv_trunk6.hip (the head conv of an f16mx6 ctx)
};
hipError_t launch_trunk_conv(TrunkArgs a, hipStream_t s);
hipError_t launch_trunk_convT(TrunkArgs a, hipStream_t s);   // conv2d_transpose_1, split-planar in and out
size_t trunk_weight_bytes();
// the residual blocks' convs in f16 + MX FP6 arithmetic on "mx6" activations (conv_trunk6.hip): a.in / a.res mx6, a.out mx6 or
// (out_split: a residual layer, not in place) split-planar; a.wpack = trunk6_pack_weights()
hipError_t launch_trunk6_conv(TrunkArgs a, int out_split, hipStream_t s);
hipError_t launch_nhwc_to_mx6(const float* x, void* out, int B, int H, int W, hipStream_t s);                // x [B,H,W,64]
hipError_t launch_mx6_to_nhwc(const void* in, float* y, int B, int H, int W, int which, hipStream_t s);      // which 0: hi + lo6, 1: x6
hipError_t launch_split_to_nhwc(const void* in, float* y, int B, int H, int W, hipStream_t s);
size_t trunk6_weight_bytes();
void trunk6_pack_weights(const float* w, unsigned char* out);   // host: Keras kernel (3, 3, 64, 64) -> trunk6_weight_bytes() bytes
// SRNet's x4 tail on a split-planar input (conv_tail.hip): conv2d_transpose_2 + ReLU + conv2d_36's tap projection and nine-tap
// sum; interior pixels of the 16 x 64 output tiles go to `img`, the tiles' border rings as partial sums to `S` (launch_gather_fix
// with th = 16 finishes them)
struct TailArgs {
    const unsigned char* in;       // [B][4 planes][H][W][64 B], H x W = the transposed conv's INPUT size
    const unsigned char* wpack;    // trunkT_entries packing of conv2d_transpose_2
    const float* bias;             // [64]
    const unsigned char* proj_w;   // ConvLayer::d_proj (bf16x3: 8 KiB)
    const float* gbias;            // [<= 3] conv2d_36's bias
    float* img;                    // [B][2H][2W][gather_cs]
    float* S;                      // [4][B][2H][2W][4] partial sums
    unsigned zero_off;             // byte offset from `in` of 64 zero bytes
    int B, H, W;
    int act;                       // A_RELU (or A_NONE) between the two convs
    int gather_cs, gather_cout, gather_act;
    int tiles_x, tiles_y;          // set by the launcher
    int dbg;                       // ablations of the micro-benchmark (SR_TAIL_DBG; 0 in production): 1 no tap sum, 2 no projection, 4 no DMA
};
hipError_t launch_tail(TailArgs a, hipStream_t s);
// FNet's 3x3 convs on split-planar activations (conv_fnet.hip): Cin = 16 npl -> Cout = 32 ncb channels, bf16x3, LeakyReLU(0.2).
// in [B][npl][H][W][64 B]; out [B][2 ncb][H][W][64 B], or its 2x2 max pooling [B][2 ncb][H/2][W/2][64 B] (pool), or fp32 NHWC
// [B][H][W][32 ncb] (out_f32).  The launcher fills the mosaic fields.
struct FConvArgs {
    const unsigned char* in;
    unsigned char* out;
    const unsigned char* wpack;   // [ncb][npl][9 taps][hi, lo][64 lanes] x 16 B (sr_api.hip fconv_entries)
    const float* bias;            // [32 ncb]
    unsigned zero_off;            // byte offset from `in` of 64 zero bytes (behind the tensor)
    unsigned dump_off;            // byte offset from `out` of the 4 KiB tail behind the output tensor
    int B, H, W;
    int npl, ncb;
    int pool, out_f32;
    int cm, cell_w, cell_h, MW, MH;   // mosaic: images per row, cell pitch, extent (set by the launcher)
    float inv_cw, inv_ch;
    int tiles_x, tiles_y;
};
hipError_t launch_fnet_conv(FConvArgs a, hipStream_t s);
size_t fnet_weight_bytes(int cin, int cout);
hipError_t launch_fnet_input_split(const float* lr_t, const float* lr_tm1, void* out, int B, int H, int W, hipStream_t s);
hipError_t launch_resize2x_split(const void* in, void* out, int B, int H, int W, int C, hipStream_t s);
// channel of k index (half h, element e) inside a 16-channel plane of the split-planar layout
__host__ __device__ constexpr int split_perm(int h, int e) { return e < 4 ? 4 * h + e : 8 + 4 * h + (e - 4); }

// host-side launchers (conv_mfma.hip, conv_tapsn.hip)
hipError_t launch_conv(int prec, int mode, int nt, int steps, const ConvArgs& a, int ncb, hipStream_t s);
hipError_t launch_conv_tapsn(int prec, int nchunk, const ConvArgs& a, hipStream_t s);
// out[y,x,co] = act(bias[co] + sum_tap P[y+ky-1, x+kx-1][tap*3+co]) (+ residual); a.in = P [B,H,W,32] fp32
hipError_t launch_tap_gather(const ConvArgs& a, hipStream_t s);
// finishes the tile-border pixels of an in-kernel gather (ConvArgs::gather_out): a = the transposed conv's arguments
hipError_t launch_gather_fix(const ConvArgs& a, hipStream_t s, int th_in = 4);   // th_in: input rows per tile (output tile 2 th_in x 64)
constexpr int GATHER_LDS = 8 * 64 * 33 * 4;   // the P records of one 8 x 64 output tile, 33-float rows
size_t conv_lds_bytes(int prec, int mode, int nt, int steps, int mt);

// element-wise / gather kernels (fused_ops.hip).  act_bf16: activations are bf16.
hipError_t launch_fnet_input(const float* lr_t, const float* lr_tm1, void* out, int act_bf16,
                             int B, int H, int W, hipStream_t s);
hipError_t launch_maxpool2(const void* in, void* out, int act_bf16, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_resize2x(const void* in, void* out, int act_bf16, int B, int H, int W, int C, hipStream_t s);
// act_bf16 = 2: the output is split-planar (conv_trunk.hip) instead of NHWC
hipError_t launch_pre_srnet(const float* flow, const float* est_prev, const float* lr_t, void* out,
                            float* hr_warp_out, int act_bf16, int B, int H, int W, hipStream_t s);
hipError_t launch_srnet_input(const float* lr_t, const float* s_s, void* out, int act_bf16,
                              int B, int H, int W, hipStream_t s);
hipError_t launch_warp3(const float* img, const float* flow, float* out, int B, int H, int W, hipStream_t s);
hipError_t launch_teco_epilogue(float* est, const float* lr_t, int B, int H, int W, hipStream_t s);

// LR synthesis of the reference's drivers: GaussianBlur 5x5 (sigma 0 table, reflect-101) + cubic x1/4, hr [N,H,W,C] -> lr [N,H/4,W/4,C]
// out[i] = in[i] / 255 (reference utils.py:57-58 normalize), n elements
hipError_t launch_u8_normalize(const unsigned char* in, float* out, size_t n, hipStream_t s);
hipError_t launch_blur_cubic_down4(const float* hr, float* lr, int N, int H, int W, int C, hipStream_t s);
hipError_t launch_cubic_up4(const float* lr, float* out, int N, int h, int w, int C, hipStream_t s);

// dense [B,rows,cols,ch] <-> zero-padded [B,rows_p,cols_p,ch] (pad mode of sr_create_padded: bottom / right padding)
hipError_t launch_pad_copy(const float* in, float* out, int B, int rows, int cols, int ch, int rows_p, int cols_p, hipStream_t s);
hipError_t launch_crop_copy(const float* in, float* out, int B, int rows, int cols, int ch, int rows_p, int cols_p, hipStream_t s);
// reference utils.py:61-63 denormalize: (x * 255).astype(uint8) (truncation); rows [0, rows) and the first row_elems_out
// elements of every row of each [HH, row_elems] image, dense output
hipError_t launch_denorm_u8(const float* est, unsigned char* out, int B, int HH, int row_elems, int rows, int row_elems_out,
                            hipStream_t s);

// generic fp32 single ops for the per-op C-ABI entry points
hipError_t launch_pad_channels(const float* in, void* out, int out_bf16, long npix, int C, int Cp, hipStream_t s);
hipError_t launch_maxpool2_generic(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_resize_generic(const float* in, float* out, int B, int H, int W, int C, int sc, hipStream_t s);
hipError_t launch_warp_generic(const float* img, const float* flow, float* out, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_s2d_generic(const float* in, float* out, int B, int H, int W, int C, int bs, hipStream_t s);

// ---- training step (train_kernels.hip): fp32 NHWC tensors with explicit channel strides ----
struct WgradArgs {
    const float* x;    // layer input [B,H,W,xcs]
    const float* dy;   // gradient of the layer output [B,H,W,ycs] (conv) or [B,2H,2W,ycs] (transposed conv)
    float* part;       // workspace of wgrad_workspace_bytes()
    float* dw;         // kernel gradient, Keras layout: conv (3,3,cin,cout), transposed conv (3,3,cout,cin)
    float* db;         // bias gradient [cout], or nullptr
    int B, H, W;       // input spatial dims
    int cin, cout, xcs, ycs, convt;
    int x3;            // 1: operands split hi+lo bf16, three bf16 MFMAs per product (ctx dtype bf16x3); 0: fp32 MFMA
    int defer_reduce = 0;   // 1: leave the slice partials in `part` (a workspace of this launch's own) for launch_wgrad_reduce_batch
};
size_t wgrad_workspace_bytes(int B, int H, int W, int cin, int cout, int convt);
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s);
// dw / db of every deferred launch in `list` from its partials, in one launch per 48 layers
hipError_t launch_wgrad_reduce_batch(const WgradArgs* list, int n, hipStream_t s);
// data gradient of Conv2DTranspose 3x3 s2: dy [B,2H,2W,ycs] (cout channels, a multiple of 64), w (3,3,cout,cin), dx [B,H,W,xcs];
// relu_y (nullable, laid out like dx): the output of the ReLU that fed the layer, dx is zeroed where it is not positive
hipError_t launch_dgrad_s2(const float* dy, const float* w, float* dx, const float* relu_y, int B, int H, int W, int cin,
                           int cout, int ycs, int xcs, hipStream_t s);
// bf16x3 variant: wpack = bf16x3 forward-conv packing of the transposed conv's kernel read as HWIO (cin' = cout, cout' = cin)
hipError_t launch_dgrad_s2_x3(const float* dy, const void* wpack, int nt, int wstride_cb, float* dx, const float* relu_y, int B,
                              int H, int W, int cin, int cout, int ycs, int xcs, hipStream_t s);
// data gradient of a 3x3 conv with <= 3 outputs (w HWIO), activation derivative of the input's producer folded in (y nullable)
hipError_t launch_dgrad_few(const float* g, const float* w, float* dx, const float* y, int act, int B, int H, int W, int cin,
                            int cout, int gcs, int xcs, hipStream_t s);
hipError_t launch_act_bwd(float* g, const float* y, int act, long npix, int C, int gcs, int ycs, hipStream_t s);
// act / (ym, act): optional ReLU / LeakyReLU derivative of the tensor the gradient lands on, folded into the store
hipError_t launch_maxpool2_bwd(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int act, hipStream_t s);
hipError_t launch_resize_bwd(const float* gy, float* gx, int B, int H, int W, int C, int sc, int ocs, int accumulate,
                             const float* ym, int act, hipStream_t s);
hipError_t launch_warp_bwd_flow(const float* img, const float* flow, const float* g, float* gflow, int B, int H, int W,
                                int C, int gfs, int accumulate, hipStream_t s);
hipError_t launch_hr_warp_bwd(const float* est_prev, const float* flow, const float* g_sin, float* g_fhr, int B, int H, int W,
                              int gcs, hipStream_t s);
size_t mse_workspace_bytes(long npix, int gcs);
hipError_t launch_mse_grad(const float* pred, const float* target, float* g, double* partial, float* loss_out, long npix, int C,
                           int gcs, const float* clip_src, int tanh_out, hipStream_t s);
hipError_t launch_adam(float* p, float* m, float* v, const float* g, long n, float lr_t, float b1, float b2, float eps,
                       float gscale, hipStream_t s);
// ents: srh::PackEnt {int32 src, uint32 dst}[n]
hipError_t launch_repack(const float* kernel, const void* ents, int n, void* out, int prec, hipStream_t s);
// batched form: job q re-packs params[src + ents[i].src] into out (ents == nullptr: copies n floats, a bias)
struct RepackJob {
    const void* ents;
    void* out;
    int n, src, block0, pad_;
};
struct RepackBatch {
    static constexpr int MAXJ = 60;   // 60 x 32 B of kernel arguments
    RepackJob j[MAXJ];
    int njobs;
};
hipError_t launch_repack_batch(const float* params, RepackBatch& b, int prec, hipStream_t s);

}  // namespace sr
