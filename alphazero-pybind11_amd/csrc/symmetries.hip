// Training-sample symmetries on the device ("next" row 1).
// Replaces GameState::symmetries(PlayHistory): Connect4GS (connect4_gs.cc:151-170) and the Tafl
// family's eightSym (tafl_helper.h:139-149, tawlbwrdd_gs.cc:455-458).  A symmetry is a pure
// permutation of the sample's cells and policy entries, so the kernel is one gather per output
// element (HBM-bound: 4 B read + 4 B written per element, fully coalesced writes); the source index
// is computed from the output index, no tables.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/azmi.h"

namespace {
thread_local std::string g_sym_err;
int fail(const char* what, hipError_t e) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  g_sym_err = buf;
  return AZMI_ERR_NO_DEVICE;
}

struct SymGeo {
  uint32_t kind;      // 0: column mirror with one policy entry per column; 1: dihedral group of a square tafl board;
                      // 2: StarGambitUnifiedGS {base, NW-axis mirror} on the 13 x 13 canvas (star_gambit_gs.cc:2623-2727)
  uint32_t C, H, W, M, NV, NS;
};

// source cell of output cell (h, w) under symmetry s
__device__ __forceinline__ void src_cell(const SymGeo& g, uint32_t s, uint32_t& h, uint32_t& w) {
  if (g.kind == 0) {
    if (s) w = g.W - 1 - w;                       // connect4_gs.cc:160
    return;
  }
  if (s >> 2) w = g.W - 1 - w;                    // mirrorWidth of the rotated sample, tafl_helper.h:33
  for (uint32_t r = s & 3; r; --r) {              // rot90Clockwise: out(h,w) = base(H-1-w, h), tafl_helper.h:77
    uint32_t nh = g.H - 1 - w;
    w = h; h = nh;
  }
}

// ---- StarGambit: the mirror about the NW axis, (q, r) -> (-q, r + q), i.e. (row, col) -> (12 - row, row + col - 6); the map is
// its own inverse, so the reference's scatter (star_gambit_gs.cc:2640-2651) is this gather.  Heading planes 9-14 are permuted
// by MIRROR_DIRECTION_MAP {4,3,2,1,0,5}, cannon planes 17-21 by {0,2,1,4,3}, action slots by SLOT_MAP {0,2,1,4,3,5,7,6,9,8}
// (star_gambit_gs.h:472-481), deploy facings by MIRROR_DIRECTION_MAP (fighters, cruisers) / DEPLOY_MIRROR_D {3,2,1,0,5,4}.
__device__ __forceinline__ uint32_t sg_dir_map(uint32_t d) { return d == 5 ? 5u : 4u - d; }
__device__ __forceinline__ uint32_t sg_lr_swap(uint32_t x) { return x == 0 ? 0u : ((x - 1u) ^ 1u) + 1u; }     // {0,2,1,4,3}
__device__ __forceinline__ uint32_t sg_slot_map(uint32_t s) { return s < 5 ? sg_lr_swap(s) : 5u + sg_lr_swap(s - 5u); }
__device__ __forceinline__ bool sg_hex(int row, int col) {     // is_unified_hex, star_gambit_gs.cc:2365-2371
  const int q = row - 6, r = col - 6, t = q + r;
  return q >= -6 && q <= 6 && r >= -6 && r <= 6 && t >= -6 && t <= 6;
}
// source element of output plane c, cell (h, w); false = the output stays 0 (no source maps there)
__device__ __forceinline__ bool sg_src_cell(uint32_t& c, uint32_t& h, uint32_t& w) {
  const int col = static_cast<int>(w) + static_cast<int>(h) - 6;
  if (col < 0 || col >= 13) return false;
  h = 12u - h; w = static_cast<uint32_t>(col);
  if (c >= 9 && c <= 14) c = 9u + sg_dir_map(c - 9u);
  else if (c >= 17 && c <= 21) c = 17u + sg_lr_swap(c - 17u);
  return true;
}
__device__ __forceinline__ uint32_t sg_src_move(uint32_t m) {
  if (m < 1690u) {
    const uint32_t slot = m % 10u, pos = m / 10u;
    const int row = static_cast<int>(pos / 13u), col = static_cast<int>(pos % 13u);
    if (!sg_hex(row, col)) return m;                                   // cells off the hex board are copied as they are, :2697-2700
    return static_cast<uint32_t>((12 - row) * 13 + (row + col - 6)) * 10u + sg_slot_map(slot);
  }
  if (m < 1708u) {
    const uint32_t d = m - 1690u, t = d / 6u, f = d % 6u;
    return 1690u + t * 6u + (t == 2 ? (f < 4 ? 3u - f : 9u - f) : sg_dir_map(f));
  }
  return m;
}

// source policy index of output policy index m under symmetry s
__device__ __forceinline__ uint32_t src_move(const SymGeo& g, uint32_t s, uint32_t m) {
  if (g.kind == 2) return s ? sg_src_move(m) : m;
  if (g.kind == 0) return s ? g.W - 1 - m : m;    // connect4_gs.cc:166
  const uint32_t span = g.W + g.H;                // policyLocation, tafl_helper.h:7-14
  uint32_t loc = m % span, sq = m / span;
  uint32_t h = sq / g.W, w = sq % g.W;
  bool hm = loc >= g.W;
  if (hm) loc -= g.W;
  if (s >> 2) {                                   // tafl_helper.h:41-48
    w = g.W - 1 - w;
    if (!hm) loc = g.W - 1 - loc;
  }
  for (uint32_t r = s & 3; r; --r) {              // tafl_helper.h:92-124
    uint32_t nh = g.H - 1 - w;
    w = h; h = nh;
    if (!hm) loc = g.W - 1 - loc;                 // row-slide to x  <- column-slide to W-1-x
    hm = !hm;                                     // column-slide to y <- row-slide to y
  }
  return (h * g.W + w) * span + (hm ? g.W : 0) + loc;
}

__global__ void k_symmetries(SymGeo g, uint32_t count, const float* __restrict__ canon, const float* __restrict__ v,
                             const float* __restrict__ pi, float* __restrict__ out_canon, float* __restrict__ out_v,
                             float* __restrict__ out_pi) {
  const uint64_t cells = uint64_t(g.C) * g.H * g.W;
  const uint64_t n_c = uint64_t(count) * g.NS * cells, n_p = uint64_t(count) * g.NS * g.M, n_v = uint64_t(count) * g.NS * g.NV;
  const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
  for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_c + n_p + n_v; i += stride) {
    if (i < n_c) {
      uint64_t smp = i / (g.NS * cells);
      uint32_t rem = uint32_t(i % (g.NS * cells));
      uint32_t s = rem / uint32_t(cells), e = rem % uint32_t(cells);
      uint32_t c = e / (g.H * g.W), h = (e / g.W) % g.H, w = e % g.W;
      bool has_src = true;
      if (g.kind == 2) { if (s) has_src = sg_src_cell(c, h, w); }
      else src_cell(g, s, h, w);
      out_canon[i] = has_src ? canon[smp * cells + (uint64_t(c) * g.H + h) * g.W + w] : 0.0f;
    } else if (i < n_c + n_p) {
      uint64_t j = i - n_c;
      uint64_t smp = j / (uint64_t(g.NS) * g.M);
      uint32_t rem = uint32_t(j % (uint64_t(g.NS) * g.M));
      uint32_t s = rem / g.M, m = rem % g.M;
      out_pi[j] = pi[smp * g.M + src_move(g, s, m)];
    } else {
      uint64_t j = i - n_c - n_p;
      uint64_t smp = j / (uint64_t(g.NS) * g.NV);
      out_v[j] = v[smp * g.NV + j % g.NV];          // v carried through unchanged (tafl_helper.h:28,71)
    }
  }
}

int run(const SymGeo& g, int device, uint32_t count, const float* canon, const float* v, const float* pi, float* out_canon,
        float* out_v, float* out_pi, int host_buffers, void* stream) {
  if (count == 0) return AZMI_OK;
  if (!canon || !v || !pi || !out_canon || !out_v || !out_pi) {
    g_sym_err = "azmi_symmetries: null buffer";
    return AZMI_ERR_INVALID;
  }
  hipError_t e;
  if (device >= 0 && (e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t cells = size_t(g.C) * g.H * g.W;
  const size_t nc = size_t(count) * cells, np = size_t(count) * g.M, nv = size_t(count) * g.NV;
  const size_t total_out = (nc + np + nv) * g.NS;
  const float *d_c = canon, *d_v = v, *d_p = pi;
  float *d_oc = out_canon, *d_ov = out_v, *d_op = out_pi, *stage = nullptr;
  if (host_buffers) {                               // numpy path: stage through one HBM block
    if ((e = hipMalloc(&stage, (nc + np + nv) * (1 + g.NS) * sizeof(float))) != hipSuccess) return fail("hipMalloc", e);
    float* p = stage;
    auto up = [&](const float* src, size_t n) { float* d = p; p += n; (void)hipMemcpyAsync(d, src, n * sizeof(float), hipMemcpyHostToDevice, st); return d; };
    d_c = up(canon, nc); d_v = up(v, nv); d_p = up(pi, np);
    d_oc = p; p += nc * g.NS; d_ov = p; p += nv * g.NS; d_op = p;
  }
  const uint32_t threads = 256;
  const uint64_t want = (total_out + threads - 1) / threads;
  const uint32_t blocks = uint32_t(want < 1 ? 1 : (want > 256u * 32u ? 256u * 32u : want));   // <= 32 blocks per CU, grid-stride beyond
  k_symmetries<<<blocks, threads, 0, st>>>(g, count, d_c, d_v, d_p, d_oc, d_ov, d_op);
  if ((e = hipGetLastError()) != hipSuccess) { if (stage) (void)hipFree(stage); return fail("k_symmetries", e); }
  if (host_buffers) {
    (void)hipMemcpyAsync(out_canon, d_oc, nc * g.NS * sizeof(float), hipMemcpyDeviceToHost, st);
    (void)hipMemcpyAsync(out_v, d_ov, nv * g.NS * sizeof(float), hipMemcpyDeviceToHost, st);
    (void)hipMemcpyAsync(out_pi, d_op, np * g.NS * sizeof(float), hipMemcpyDeviceToHost, st);
    e = hipStreamSynchronize(st);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail("azmi_symmetries sync", e);
  }
  return AZMI_OK;
}
}  // namespace

extern "C" {

const char* azmi_symmetries_last_error(void) { return g_sym_err.c_str(); }

uint32_t azmi_num_symmetries(int game) {
  return (game == AZMI_GAME_CONNECT4 || game == AZMI_GAME_STARGAMBIT) ? 2u
         : (game == AZMI_GAME_TAWLBWRDD || game == AZMI_GAME_BRANDUBH || game == AZMI_GAME_OPENTAFL) ? 8u : 0u;
}

int azmi_symmetries(int game, int device, uint32_t count, const float* canon, const float* v, const float* pi,
                    float* out_canon, float* out_v, float* out_pi, int host_buffers, void* stream) {
  SymGeo g;
  if (game == AZMI_GAME_CONNECT4) g = SymGeo{0, 4, 6, 7, 7, 3, 2};
  else if (game == AZMI_GAME_TAWLBWRDD) g = SymGeo{1, 7, 11, 11, 11 * 11 * 22, 3, 8};
  else if (game == AZMI_GAME_BRANDUBH) g = SymGeo{1, 7, 7, 7, 7 * 7 * 14, 3, 8};
  else if (game == AZMI_GAME_OPENTAFL) g = SymGeo{1, 8, 11, 11, 11 * 11 * 22, 3, 8};
  else if (game == AZMI_GAME_STARGAMBIT) g = SymGeo{2, 36, 13, 13, 1709, 3, 2};
  else { g_sym_err = "azmi_symmetries: unknown game"; return AZMI_ERR_INVALID; }
  return run(g, device, count, canon, v, pi, out_canon, out_v, out_pi, host_buffers, stream);
}

int azmi_tafl_symmetries(uint32_t board, uint32_t channels, uint32_t num_values, int device, uint32_t count,
                         const float* canon, const float* v, const float* pi, float* out_canon, float* out_v,
                         float* out_pi, int host_buffers, void* stream) {
  if (board == 0 || board > 32 || channels == 0 || num_values == 0) { g_sym_err = "azmi_tafl_symmetries: bad geometry"; return AZMI_ERR_INVALID; }
  SymGeo g{1, channels, board, board, board * board * 2 * board, num_values, 8};
  return run(g, device, count, canon, v, pi, out_canon, out_v, out_pi, host_buffers, stream);
}

}  // extern "C"
