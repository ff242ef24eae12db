// Launch wrappers for the gfx950 kernels of the batched NN evaluation path.
// Activations live in HBM as  act[board][square 0..63][channel]  (channel contiguous, "NHWC"), element type
// T = _Float16 (Precision float16: f16 MFMA operands, f32 accumulate) or float (Precision float32: exact f32 MFMA).
// square = h*8 + w of the reference's NCHW planes (engine/src/environments/chess_related/inputrepresentation.cpp:33-46).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace cra {

typedef _Float16 half_t;

constexpr int kSquares = 64;

// ---- packed-weight geometry shared by host packer and kernels -------------------------------------------------
// Dense conv weights are stored per (cout-tile of 16, k-slab of 32) as one MFMA A-fragment image:
//   wpk[((ct * nslab + s) * 64 + lane) * 8 + j] = W[co = ct*16 + (lane & 15)][k = s*32 + (lane >> 4)*8 + j]
// with k ordered [tap = kh*KS+kw][ci] (ci padded to a multiple of 32), so one wave-wide 16-B load per lane
// fetches a whole fragment as 1 KiB contiguous (f16) / 2 KiB (f32).
inline int64_t packed_weight_elems(int cout_pad, int k_total) { return int64_t(cout_pad) * k_total; }

struct ConvArgs {
    const void* x;        // [B][64][cin]   T
    const void* wpk;      // packed weights T (Precision float16x3: the hi halves, f16)
    const void* wpk_lo;   // Precision float16x3 only: the lo halves (same fragment image), else nullptr
    const float* bias;    // [cout_pad]
    const void* resid;    // optional [B][64][cout_ld] T (added before the activation)
    void* out;            // T [B][64][cout_ld]   or   float [B][cout_real*64] (policy-map, channel-major)
    int batch;
    int cin;              // multiple of 32
    int cout_pad;         // multiple of 16
    int cout_real;
    int cout_ld;          // row pitch of `out` in elements (NHWC mode)
    int ks;               // 1 or 3
    int relu;             // 0 none, 1 after the residual add, 2 before it (ClassicalResidualBlock)
    int out_policy_f32;   // 1: write float logits channel-major
    int out_flat;         // 1: write T channel-major flat, out[b*flat_pitch + co*64 + sq] (the value head's .view(-1, nb_flatten))
    int flat_pitch;
    int out_rows_f32;     // 1: write float row-major out[(b*64 + sq) * cout_real + co] for rows < rows_valid (an FC over the batch)
    int rows_valid;
    // Precision float16x3, with out_policy_f32 and the whole cout range in ONE workgroup: softmax over the board's cout_real * 64 logits
    // in the same launch, probabilities to softmax_out[b * cout_real * 64 + ...] (policy_softmax); the logits go to `out` only if it is set
    float* softmax_out;
    // Precision float16x3: x is not an activation tile but the NCHW input planes [B][planes_c][64] (float); channels planes_c ... cin - 1
    // read as zeros (the stem conv does the planes -> NHWC transform while it stages the board)
    const float* planes;
    int planes_c;
    // Precision float16p8 (x3.hip: conv3x3_p8_kernel; ks = 3, cin a multiple of 128): wpk = f16 image of w * 2^p, wpk_lo = the 8-bit image of the
    // cross terms (rise_net.hip: pack_dense_p8), acc_scale = 2^-p
    int p8;
    float acc_scale;
    // float16p8, the policy head of a policy-map net in one launch (conv3x3_p8_chain_kernel): the conv 3x3 256 -> 256 + BN + ReLU in FRONT of this
    // conv -- its weight images, bias [256] and 2^-p; x is then that conv's input
    const void *pre_wpk, *pre_wpk_lo;
    const float* pre_bias;
    float pre_acc_scale;
    int dev;              // development (CRA_X3_CONV_DEV): 1 = every wave leaves the kernel behind one last barrier, 2 = waves without a cout
                          // tile request no weight fragments
    int few_boards;       // the net was made for a small batch (board-split forward): the couts of a wide layer go over several workgroups per
                          // board -- 1: two of 128 couts, 2: four of 64 (RiseNet::DevSwitches::small_conv_split)
};

template <typename T> void launch_conv_gemm(const ConvArgs& a, hipStream_t s);

// Fused mobile-bottleneck block (_BottlekneckResidualBlock, builder_util.py:437-475):
//   y = x + BN3(conv1x1_project(ReLU(BN2(dw_kxk(ReLU(BN1(conv1x1_expand(x))))))))     (x already SE-scaled if the block has SE)
// One workgroup per board; the C_op-wide intermediate never leaves the CU (LDS tiles of 64 channels), the 256 x 64
// project accumulator lives in registers.  Algorithmic HBM bytes per board: 2 * 64*C*sizeof(T) (x in, y out); weights
// (2*C*C_op*sizeof(T) + small) stream from L2.
struct BlockArgs {
    const void* x;        // [B][64][C] T
    void* y;              // [B][64][C] T
    const void* w1pk;     // expand  weights packed (cout = cop_pad, k = C)
    const float* b1;      // [cop_pad]
    const float* wdw;     // [ks*ks][cop_pad] float (folded), zero padded
    const float* b2;      // [cop_pad]
    const void* w3pk;     // project weights packed (cout = C, k = cop_pad)
    const void* w1pk_lo;  // Precision float16x3 only (x3.hip): lo halves of the split expand / project weights, w1pk / w3pk hold the hi halves
    const void* w3pk_lo;
    const float* b3;      // [C]
    int batch, C, cop_pad, ks;
    const float* gate;    // optional [B][C]: SE gate multiplied into x while the tile is loaded (residual uses the gated x)
    float* pool_out;      // optional [B][C]: sum over the 64 squares of y (feeds the NEXT block's SE gate)
    const float* dwpk;    // 3x3 only: [cop_pad][12] = 9 folded taps, BN1 bias, BN2 bias, 0 (one 48-byte record per channel); float16x3: X3TowerBlock's tile layout
};
template <typename T> void launch_block(const BlockArgs& a, hipStream_t s);
// Precision float16x3 (x3.hip): split-operand f16 MFMAs (a = hi + lo; hi*hi + hi*lo + lo*hi, f32 accumulate) on float activations.
// The conv GEMM covers every dense layer of every net family; the fused block covers the 3x3 bottleneck blocks of 256-channel nets.
void launch_conv_gemm_x3(const ConvArgs& a, hipStream_t s);
void launch_block_x3(const BlockArgs& a, hipStream_t s);          // needs dwpk, w1pk_lo, w3pk_lo; ks == 3
void init_x3_kernel_attributes();
int block_x3_chunk_channels();
// a run of consecutive 3x3 bottleneck blocks in one launch (x3.hip: tower_x3_kernel): the residual stream stays in LDS as a hi / lo f16
// pair, the SE gate of every block but the first is computed in-kernel from float weights (the first block's gate, if any, is applied
// by the caller's SE launch)
struct X3TowerBlock {
    const void *w1pk, *w1pk_lo, *w3pk, *w3pk_lo;     // as BlockArgs
    const float* dwpk;                               // [cop_pad / 16 tiles][16 rows: taps dx = -1 (dy = -1, 0, 1), dx = 0, dx = +1, BN1 bias, BN2 bias, 5 x 0][16 channels] (rise_net.hip: pack_x3_depthwise_records)
    const float* b3;                                 // [256]
    const float* se_w1t;                             // gate matrices in THREAD order (x3.hip: x3_se_phase; rise_net.hip: pack_se_threads_f32):
    const float* se_w2t;                             //   ca_se: W1 then W2, 16 float4 loads per thread each; eca_se: se_w1t = both halves
    const float* se_b;                               // eca_se: [256]
    int cop_pad;                                     // multiple of block_x3_chunk_channels()
    int se_kind;                                     // 0 none, 1 ca_se, 2 eca_se
    // Precision float16p8 (x3.hip, tower_p8_kernel): w1pk / w3pk = f16 image of w * 2^p (p per layer), w1pk_lo / w3pk_lo = the 8-bit image of the cross
    // terms (per cout tile and 64 k: lanes' 32 bytes [e5m2((w * 2^p) - hi) for 64 k ; e5m2(hi) for the same 64 k], both times the truncation
    // compensation, bytes 0-15 in "slab" 2 J, bytes 16-31 in "slab" 2 J + 1 of the lo image's geometry; rise_net.hip: pack_dense_p8).  w1_inv = 2^-p1
    // brings the expand accumulators back in front of the BN1 bias; the residual stream runs in the project weights' scale inside a block:
    // x := (x + b3) * w3_scale, + the project sums, x := x * w3_inv (powers of two: exact)
    float w1_inv, w3_scale, w3_inv;
};
struct X3TowerArgs {
    const float* x;       // [B][64][256]
    float* y;             // [B][64][256]
    const X3TowerBlock* blocks;   // device array
    int nblocks;
    int batch;
    int p8;               // Precision float16p8 (tower_p8_kernel)
    int ks;               // float16p8: the depthwise size of every block of the run, 3 or 5 (0 = 3); a gated FIRST block has its gate computed in the launch
    int symmetric;        // development (CRA_X3_TOWER=symmetric when the net was made): float16x3's 3x3 runs on tower_x3_kernel, every wave all three phases
};
void launch_tower_x3(const X3TowerArgs& a, hipStream_t s);
// Small batches (round 6): ONE 3x3 bottleneck block per launch with G workgroups per board (x3.hip: block_x3_split_kernel).  Workgroup g of a
// board stages the whole board (every workgroup needs all 256 input channels of the expand GEMM), runs the chunks [g n / G, (g + 1) n / G) of
// the block's n = C_op / 128 chunks through expand -> depthwise -> project (float16x3 arithmetic, x3_chunks) and writes its PARTIAL project
// sums -- workgroup 0 also the residual x + b3 -- as a float image of its own, [B][G][64][256]; the next launch adds the G images of a board
// in the order of their index while it stages (the same bits in every workgroup and on every launch: no atomics, nothing depends on the
// order in which workgroups arrive).  Two image sets alternate.  launch_x3_split_finish adds the last block's images into the float stream
// the heads read.  One board per workgroup costs a batch of ONE the whole tower's latency on one CU (0.46 ms for RISEv2-19); this form
// spreads a board's block over up to n CUs.  (Built first with 64-bit fixed-point atomic adds into one sum per board: order-free as well,
// but device-scope atomics execute at the memory side on this chip -- 4.5 us per block at batch 1, 1 ms per forward at batch 32;
// profiles/r06/b_*.)
struct X3SplitArgs {
    X3TowerBlock blk;         // this launch's block (SE gate, if any, computed by every workgroup from the staged board)
    const float* x_parts;     // [B][gin][64][256]: the stream = the sum of these images (the run's first block: gin = 1, the float stream itself)
    float* y_parts;           // [B][G][64][256]
    int gin;                  // 1 ... 16
    int batch, G;             // 1 <= G <= min(cop_pad / 128, 16)
    // A gated block needs the board's channel means before it can use the board.  Means are linear: the launch before it leaves, per
    // workgroup, the channel sums of the image it wrote ([B][G][256], pool_out), and the gated block adds them up (pool_in, gin of them per
    // board) and has its gate BEFORE it stages the board -- the gate is then applied while staging.  nullptr: the gate phase on the staged
    // tiles (a run's first block; CRA_X3_SPLIT_DEV bit 8).
    const float* pool_in;
    float* pool_out;
    int dev;                  // development (CRA_X3_SPLIT_DEV when the net was made; timing switches, x3.hip)
};
void launch_block_x3_split(const X3SplitArgs& a, hipStream_t s);
void launch_x3_split_finish(const float* parts, int gin, float* y, int batch, hipStream_t s);
template <typename T> void init_block_kernel_attributes();
template <typename T> int block_chunk_channels();   // C_op must be padded to a multiple of this

// Residual tower: a run of consecutive 3x3 bottleneck blocks in one launch, one workgroup per board, residual stream
// resident in LDS, SE gates computed in-kernel (tower.hip).  f16 only, C = 256, C_op padded to whole chunks of 128.
// The conv weights of the whole run are packed into per-wave streams in consumption order (rise_net.hip):
//   wstream  4 matrix waves x fragments of 64 lanes x 8 halves = v_mfma_f32_32x32x16_f16 A operands (lane l, element j:
//            row l%32, k (l/32)*8 + j).  Per block, per interval k = -1..n: 16 expand fragments of chunk k+1 ([k-step of 16],
//            rows = channels chunk*128 + w*32 + row), then 16 project fragments of chunk k-1 ([k-step of 16][row tile of 32],
//            rows = couts w*64 + rt*32 + row, k = tower K position); kTowerWindow zero fragments at the very end
//   bstream  4 matrix waves x per chunk [lane/32][element v] BN1 bias of row (v%4) + 8*(v/4) + 4*(lane/32)   (32 floats)
//   pstream  4 vector waves x per chunk 2 KiB of half2 for K positions w*32 + lg*8 + pi*2 + {0,1}; entries = k*k folded taps then the BN2
//            bias (k = 3: 0..8, 9; k = 5: 0..24, 25), rest zero; one chunk of padding at the end
//              k = 5: [32 entries][lane group lg][4 channel pairs]
//              k = 3: [10 entries][lane group lg][file variant][4 channel pairs]: variant 0 = for squares on file a (dx = -1 taps zero),
//                     1 = files b..g, 2 = file h (dx = +1 taps zero)
constexpr int kTowerWindow = 16;
struct TowerBlockDesc {
    const float* b3;      // [256] BN3 bias (Precision fp8: divided by s3)
    const float* s3;      // Precision fp8: [256] power-of-two scale of the project weights per cout (y = x + s3 * acc); else nullptr
    const void* se_w1;    // f16 pairs in thread order (rise_net.hip: pack_se_threads; tower.hip: se_phase): ca_se W1, or the eca_se centre-tap
                          // matrix in two halves; or nullptr
    const void* se_w2;    // f16 pairs in thread order: ca_se W2
    const float* se_b;    // eca_se bias [256]
    int cop_pad;          // multiple of 128
    int ks;               // depthwise kernel size: 3 or 5
    int se_kind;          // 0 none, 1 ca_se, 2 eca_se: gate applied to this block's input (from the previous block's output sums)
    // Precision int8 (TowerArgs::fp8 == 2; tower.hip Q = 2): the calibrated activation steps of this block as their reciprocals (f16
    // numbers): qx_inv for the (gated) stream in front of it (+-127), qt_inv for its depthwise output (0 ... 255); escale = what the
    // expand accumulators are multiplied with on their way to f16 (2^-7).  b3 then holds INT32 bit patterns -- the BN3 bias in units of
    // the project accumulator plus 128 x the row's weight sum (the depthwise output is stored as u - 128) -- and s3 the value of such a
    // unit per cout; the bias stream holds the BN1 biases as int32 bit patterns likewise.
    float qx_inv, qt_inv, escale;
};
struct TowerArgs {
    const void* x;        // [B][64][256] f16
    void* y;              // [B][64][256] f16
    const TowerBlockDesc* blocks;   // device array
    const void* wstream;
    const float* bstream;
    const void* pstream;
    long long wstream_wave_frags, bstream_wave_floats, pstream_wave_bytes;    // per-wave stream lengths
    int nblocks;
    int batch;
    const float* gate_in; // optional [B][256]: SE gate of blocks[0] computed by a previous launch (blocks[0].se_kind is ignored)
    float* pool_out;      // optional [B][256]: sum over the 64 squares of y (feeds an SE gate computed by a later launch)
    unsigned long long* trace;   // development: s_memtime stamps of workgroup 0 (CRA_TOWER_TRACE), [wave 0 | wave 4][256]
    int fp8;              // 2: Precision int8 (same streams, int8 bytes, v_mfma_i32_32x32x32_i8; TowerBlockDesc).  1: Precision fp8: e4m3 GEMM operands (v_mfma_f32_32x32x64_f8f6f4).  wstream = bytes; a matrix wave's region holds TWO
                          // streams of 1 KiB loads: [0, wstream_e_frags) expand, per chunk 8 loads [k-step of 64][half][lane][16 B] (lane l = row
                          // l%32, bytes k = ks*64 + (l/32)*32 + half*16 + t), then project, per chunk 8 loads [k-step][row tile][half][lane][16 B];
                          // each closed by 8 loads of zeros.  Weights divided by a power of two per row (max |w_q| in [1, 2)): the expand
                          // scale is folded into the depthwise weights, bstream = b1 / s1, b3 = b3 / s3
    long long wstream_e_frags;
    void* block_dump;     // test hook (mi_net_block_dump), else nullptr: f16 [nblocks + 1][B][64][256] -- tile 0 = the residual stream as the
                          // tower received it, tile i + 1 = the stream behind block i (before the next block's SE gate scales it)
};
void launch_tower(const TowerArgs& a, hipStream_t s);
void init_tower_kernel_attributes();
size_t tower_lds_bytes();
// K position of the tower's project GEMM (= position inside the t1 / t2 tiles) -> C_op channel.  Chunks of 128; wave w's
// 32 positions hold its 32 expand rows in the order a 32x32 MFMA result leaves them in a lane: position p <-> row
// (p%4) + 8*((p%16)/4) + 4*(p/16), so that a lane's 16 results are 16 consecutive positions.
inline int tower_row_of_position(int p) { return (p % 4) + 8 * ((p % 16) / 4) + 4 * (p / 16); }
inline int tower_k_channel(int kpos) {
    const int chunk = kpos / 128, wq = (kpos % 128) / 32, p = kpos % 32;
    return chunk * 128 + wq * 32 + tower_row_of_position(p);
}

// Stem in one launch, one workgroup per board (stem.hip): fp32 NCHW planes -> conv3x3 + BN + ReLU -> f16 NHWC.
struct StemArgs {
    const float* planes;  // [B][cin][64] fp32 NCHW (the NeuralNetAPI::predict input)
    void* x;              // [B][64][256] f16
    const void* stem_w;   // 8 waves x [9 taps][cin_pad/16 k-steps] A fragments (32x32x16: lane l, element j: cout 32*wave + l%32,
                          // cin k-step*16 + (l/32)*8 + j), then 16 zero fragments
    const float* stem_b;  // [wave][lane/32][16] folded BN bias, rows (v%4) + 8*(v/4) + 4*(lane/32)
    long long stem_wave_frags;
    int cin, cin_pad;     // cin_pad: multiple of 16 in [48, 96]
    int batch;
    // search lanes: instead of reading `planes`, the kernel builds its input tile from 192-byte board descriptors (the same plane_value()
    // the stand-alone builder uses; normalised planes), boards >= n_valid read zeros.  nullptr = planes path.
    const void* descs;
    int layout, n_valid;
};
void launch_stem(const StemArgs& a, hipStream_t s);

// Policy + value head in one launch, one workgroup per board (head.hip).  f16, C = 256, value head 8 channels / 512 flat.
//   s1  8 waves x [9 taps][16 k-steps] A fragments of policy conv 1 (rows = couts 32*wave + row), then 16 fragments of the
//       value head's 1x1 conv for wave 0 (rows 0..7; zeros for the other waves), then 16 zero fragments
//   b1  [wave][lane/32][16] folded BN bias of policy conv 1, rows (v%4) + 8*(v/4) + 4*(lane/32)
//   s2  8 waves x 18 (tap, k-step) units x 3 row tiles of policy conv 2 (unit u = 18*wave + i: tap u/16, k-step u%16; K position
//       -> conv-1 channel as in tower_k_channel), then 9 zero fragments
struct HeadArgs {
    const void* x;            // [B][64][256] f16
    float* logits;            // [B][cp*64] policy_out (pre-softmax), or nullptr: not written
    float* probs;             // [B][cp*64] softmax, or nullptr: not written (a search lane that only takes g_out)
    float* value;             // [B]
    float* aux;               // [B][4] or nullptr
    const void* s1;
    const float* b1;
    const void* s2;
    long long s1_wave_frags, s2_wave_frags;
    const float* vconv_bias;  // [8]
    const void* fc1_w;        // tanh head: f16 pairs in thread order (head.hip, phase 4);  WDLP head: float [4][512] (wdl rows 0..2, plys)
    const float* fc1_b;       // [256]
    const float* fc2_w;       // [256]
    float fc2_b;
    float wdl_b[4];
    int cp;                   // policy channels, <= 96
    int wdlp;
    int batch;
    unsigned long long* trace;   // development: s_memtime stamps of workgroup 0, wave 0 (CRA_TOWER_TRACE)
    // search lanes: board b < g_n_valid also writes probs[g_idx[b * g_stride + j]] to g_out[b * g_stride + j], j < g_cnt[b] (what the
    // gather kernel does behind a three-launch lane step); nullptr = off
    const uint16_t* g_idx;
    const uint32_t* g_cnt;
    float* g_out;
    int g_stride, g_n_valid;
};
void launch_head(const HeadArgs& a, hipStream_t s);
void init_head_kernel_attributes();
size_t head_lds_bytes();

// Stem + tower + head of a board in one launch (forward.hip): the board tile stays in LDS across both seams.  Needs a tower launch
// that covers every block (gate_in == pool_out == nullptr); x / y of the three argument sets are not touched.
void launch_forward(const StemArgs& sa, const TowerArgs& ta, const HeadArgs& ha, hipStream_t s);
void init_forward_kernel_attributes();

// Dense residual tower in one launch, one workgroup per board (restower.hip).  f16, C = 256.
//   wstream  (8 / NR) waves x per block { conv 1: [9 taps][16 k-steps][NR cout tiles] A fragments (rows = couts
//            32*(NR*wave + rt) + row, K = input channel), conv 2: the same with K position -> conv-1 channel
//            (kpos/32)*32 + tower_row_of_position(kpos%32) }, then 16 zero fragments;  NR = cout_tiles_per_wave
//   bstream  (8 / NR) waves x per block { conv 1 BN bias [rt][lane/32][16], conv 2 BN bias [rt][lane/32][16] } in accumulator row order
struct ResTowerArgs {
    const void* x;            // [B][64][256] f16
    void* y;
    const void* wstream;
    const float* bstream;
    long long wstream_wave_frags;
    long long bstream_wave_floats;
    int nblocks;
    int relu_after_add;       // 1: ReLU(x + body(x)) (AlphaZero ResidualBlock), 0: x + ReLU(body(x)) (ClassicalResidualBlock)
    int batch;
    int boards_per_workgroup; // 1 or 2 (restower.hip: a weight fragment feeds 2 or 4 MFMAs)
    int cout_tiles_per_wave;  // 1: 8 waves x 32 couts, 2: 4 waves x 64 couts (a tile fragment feeds 1 or 2 MFMAs); fixes the stream layout
};
void launch_restower(const ResTowerArgs& a, hipStream_t s);
void init_restower_kernel_attributes();
size_t restower_lds_bytes();

// depthwise k x k (k = 3 or 5) + folded BN + ReLU.  w: [k*k][C] float, bias: [C] float
template <typename T> void launch_depthwise(const T* x, T* y, const float* w, const float* bias, int batch, int C, int ks,
                                            hipStream_t s);

// squeeze-excitation, in place on x [B][64][C].  kind 1 = ca_se: w1t [C][C/2], w2t [C/2][C] (both transposed, no bias);
// kind 2 = eca_se: w1t [C][C] transposed centre tap, b1 [C].  hard-sigmoid gate (builder_util.py:452).
// kind_flags: 1 ca_se / se, 2 eca_se, + 16 plain sigmoid; res: optional shortcut, x = relu(res + x * gate)
template <typename T> void launch_se(T* x, int kind_flags, const float* w1t, const float* w2t, const float* b1, int batch, int C,
                                     hipStream_t s, const T* res = nullptr);

// SE gate MLP on pooled sums (the squeeze already happened in the producer's epilogue): gate[b][c] = hard_sigmoid(...)
//  kind 1 (ca_se):  relu(W1 mean) -> W2 -> hard-sigmoid   w1t [C][C/2], w2t [C/2][C]
//  kind 2 (eca_se): Wc mean + b -> hard-sigmoid            w1t [C][C],   b1 [C]
// 8 boards per 1024-thread workgroup; every weight is loaded once per workgroup with all loads of a thread in flight.
void launch_se_gate(const float* pool, float* gate, int kind, const float* w1t, const float* w2t, const float* b1, int batch, int C,
                    hipStream_t s);

struct ValueHeadArgs {
    const void* x;          // [B][64][C] T
    const float* wconv;     // [cv][C] folded
    const float* bconv;     // [cv]
    const float* w1t;       // [64*cv][fc] transposed (tanh head)
    const float* b1;        // [fc]
    const float* w2;        // [fc]
    float b2;
    const float* wwdl;      // [3][64*cv]   (wdl head) or nullptr
    const float* bwdl;      // [3]
    const float* wplys;     // [64*cv]
    float bplys;
    float* value;           // [B]
    float* aux;             // [B][4] or nullptr
    int batch, C, cv, fc;
    // development (CRA_VALUE_HEAD_DEBUG, scripts/lane_divergence.py): [B][8] stage checksums of the launch -- staged board, staged conv
    // weights, conv output, FC1 partial sums, FC2 sum, value -- each added up in a fixed order: equal inputs give equal bits
    float* dbg;
    int lds_pad;            // < 0 (default): only the LDS the kernel uses; >= 0 (development, CRA_VALUE_HEAD_LDS_PAD=0): round 4's fence, 144 KB
                            // requested so that the workgroup has its compute unit to itself
    int variant;            // development (CRA_VALUE_HEAD_VARIANT): 1 = FC1 partial sums in LDS of their own (not over the dead board tile),
                            // 2 = FC1 accumulators pinned per step (no packed f32 FMAs), 4 = s_waitcnt vmcnt(0) behind every group of 32 weight
                            // loads, 8 = weight loads non-temporal, 16 = the PROBE instantiation (kernels.hip: sums from the registers, read
                            // back from LDS, checksums of the loaded words, HW_ID per wave), 32 = FC1 as plain fmaf (v_pk_fma_f32 in a build with packed f32 ops) instead of the v_fmac_f32 asm,
                            // 64 = four waves per board (the form up to round 6's set D; the product form runs eight: 22.5 -> 17 us at batch 256)
};
template <typename T> void prepare_value_head(const ValueHeadArgs& a);   // once per net: LDS allowance of the kernel
size_t value_head_lds_bytes(const ValueHeadArgs& a);
template <typename T> void launch_value_head(const ValueHeadArgs& a, hipStream_t s);

// Small batches (nets made for at most 64 boards, Precision float16x3 / float16p8): the second policy conv with its softmax and the value
// head as the two roles of ONE launch, side by side (x3.hip: heads_small_kernel)
struct HeadsSmallArgs {
    ConvArgs conv;
    ValueHeadArgs vh;
};
bool heads_small_fits(const ConvArgs& c, const ValueHeadArgs& v);
void launch_heads_small(const HeadsSmallArgs& a, hipStream_t s);

// last stage of the value head, one wave per board.
//  tanh head : value = tanh(b2 + dot(w2, h[b]))            h: [B][fc] T (FC1 + ReLU output of the conv_gemm "FC" launch)
//  WDLP head : wdl = W_wdl flat + b, plys = sigmoid(w_plys flat + b); value = -softmax(wdl)[0] + softmax(wdl)[2]; aux = [wdl, plys]
struct ValueFinalArgs {
    const void* in;         // T [B][n]  (n = fc for the tanh head, 64*cv for WDLP)
    int n;
    const float* w;         // [n] (tanh head)  or  [4][n] = wdl rows 0..2, plys row (WDLP)
    float b[4];
    int wdlp;
    float* value;           // [B]
    float* aux;             // [B][4] or nullptr
    int batch;
};
template <typename T> void launch_value_final(const ValueFinalArgs& a, hipStream_t s);

// out[s][j] = probs[s][idx[s][j]] for j < cnt[s], s < n_slots (rows of `stride` entries): the priors of the legal moves of each new
// search node, gathered next to the network output (set_probabilities_for_moves, node.cpp:961-979, reads exactly these entries).
// idx / cnt / out / value_out / aux_out may be pinned host memory (read and written in place over PCIe); the same launch copies
// the batch's values (and aux) out.
void launch_gather_probs(const float* probs, int nb_policy, const uint16_t* idx, const uint32_t* cnt, int stride, int n_slots, float* out,
                         const float* value_dev, float* value_out, int batch, const float* aux_dev, float* aux_out, hipStream_t s);

// row softmax over n logits per board (tensorrtapi.cpp:378-392 appends exactly this to policy_out)
void launch_softmax(const float* logits, float* probs, int batch, int n, hipStream_t s);

// float NCHW planes [B][C][64] (the NeuralNetAPI::predict input) -> T [B][64][cpad], zero channel padding
template <typename T> void launch_planes_to_act(const float* planes, T* act, int batch, int C, int cpad, hipStream_t s);

}  // namespace cra
