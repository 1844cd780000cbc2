// lp_renderer_loop.hip -- host side of the layer-looped bf16x3 MFMA family of the Renderer + its forward kernels and the
// deep (up to 4 / 4 / 4 layers) and two-block (hidden 64 / 64 grid channels) backward instantiations.  Device code:
// lp_renderer_loop.h; the shallow two-waves-per-SIMD backward: lp_renderer_loop_shallow.hip.
#include "lp_renderer_loop.h"

namespace lp {

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int loop_nb(int H, int C = 16) { return (H <= 32 && C <= 32) ? 1 : 2; }

// Shape family: grid-list(s) with C in {16, 32} channels below 2^31 rows (any byte size); trunk of 1..4 layers, or none with a separate colour
// grid-list; heads of 1..4 layers; every hidden width equal to H in {16, 32} -- or, on the two-block instantiation (H = 64, or
// C = 64 grid channels with any of the three widths): at most 2 trunk layers, heads of at most 2 layers, <= 4 colour channels, a
// colour grid only with C <= 32; <= 256 beyond-far samples.
bool renderer_loop_supported(const LpRendererArgs& a, const char** why) {
  *why = "";
  const int C = a.grid.channels;
  const bool tg = a.color_grid.n_grids > 0;
  if (C != 16 && C != 32 && C != 64) { *why = "grid channels not 16, 32 or 64"; return false; }
  if ((tg && a.trunk.n_layers != 0) || (!tg && a.trunk.n_layers < 1) || a.trunk.n_layers > LOOP_MAX_T || a.opacity.n_layers < 1 ||
      a.opacity.n_layers > LOOP_MAX_H + 1 || a.color.n_layers < 1 || a.color.n_layers > LOOP_MAX_H + 1) {
    *why = "layer counts outside trunk 1-4 (0 with a colour grid) / opacity 1-4 / colour 1-4";
    return false;
  }
  int H = 0;
  bool same = true;
  auto hidden = [&](int w) { if (H == 0) H = w; else same = same && (w == H); };
  for (int l = 1; l <= a.trunk.n_layers; ++l) hidden(a.trunk.dims[l]);
  for (int l = 1; l < a.opacity.n_layers; ++l) hidden(a.opacity.dims[l]);
  for (int l = 1; l < a.color.n_layers; ++l) hidden(a.color.dims[l]);
  if (H == 0) H = C;  // two-grid decoder with single-layer heads: no hidden layer at all
  if (!same) { *why = "hidden widths differ between layers"; return false; }
  if (H != 16 && H != 32 && H != 64) { *why = "hidden width other than 16 / 32 / 64"; return false; }
  if ((H == 64 || C == 64) && (a.trunk.n_layers > 2 || a.opacity.n_layers > 2 || a.color.n_layers > 2)) {
    *why = "hidden width 64 / 64 grid channels with more than 2 layers per MLP";
    return false;
  }
  if (C == 64 && tg) { *why = "64 grid channels with a separate colour grid"; return false; }
  if (a.color_chn > 32) { *why = "more than 32 colour channels"; return false; }
  if (a.color_chn > 4 && (H == 64 || C == 64)) { *why = "more than 4 colour channels with hidden width 64 / 64 grid channels"; return false; }
  if (!grid_list_rows_ok(a.grid)) { *why = "grid-list of 2^31 rows or more (or a grid slice of 2 GB or more)"; return false; }
  if (tg && !grid_list_rows_ok(a.color_grid)) { *why = "colour grid-list of 2^31 rows or more (or a grid slice of 2 GB or more)"; return false; }
  if (a.march.num_samples_inf > LOOP_N_INF) { *why = "more than 256 beyond-far samples"; return false; }
  return true;
}

static LoopParams loop_params(const LpRendererArgs& a) {
  LoopParams p = {};
  const int C = a.grid.channels;
  const bool tg = a.color_grid.n_grids > 0;
  p.n_t = a.trunk.n_layers;
  p.n_o = a.opacity.n_layers - 1;
  p.n_c = a.color.n_layers - 1;
  const int H = p.n_t > 0 ? a.trunk.dims[1] : (p.n_o > 0 ? a.opacity.dims[1] : (p.n_c > 0 ? a.color.dims[1] : C));
  const int NB = loop_nb(H, C);
  p.hid = H;
  p.hin = tg ? C : H;
  p.ho_w = p.n_o > 0 ? H : p.hin;
  p.hc_w = p.n_c > 0 ? H : p.hin;
  // small block (floats): biases of the layers, output layers of the heads, beyond-far table
  int f = 0;
  int img = 0;  // bytes, relative to the end of the small block (fixed up below)
  auto layers = [&](const LpMlp& m, int n, LoopLayer* out) {
    int64_t off = m.offset;
    // flat layout of an MLP: all weight matrices, then all biases (mlp_utils.py flatten_*)
    int64_t boff = m.offset;
    for (int l = 0; l < m.n_layers; ++l) boff += (int64_t)m.dims[l] * m.dims[l + 1];
    for (int l = 0; l < m.n_layers; ++l) {
      if (l < n) {
        out[l].w = off;
        out[l].b = boff;
        out[l].rows_in = m.dims[l];
        out[l].cols = m.dims[l + 1];
        out[l].ld = m.dims[l + 1];
        out[l].ob = (m.dims[l + 1] + 31) / 32;
        out[l].bias = f;
        f += 32 * NB;
        out[l].img = img;
        img += loop_layer_bytes(m.dims[l], m.dims[l + 1]);
      }
      off += (int64_t)m.dims[l] * m.dims[l + 1];
      boff += m.dims[l + 1];
    }
  };
  layers(a.trunk, p.n_t, p.t);
  layers(a.opacity, p.n_o, p.o);
  layers(a.color, p.n_c, p.c);
  // output layers: the last weight matrix / bias of each head
  auto last = [&](const LpMlp& m, int64_t& w, int64_t& b) {
    int64_t off = m.offset, boff = m.offset;
    for (int l = 0; l < m.n_layers; ++l) boff += (int64_t)m.dims[l] * m.dims[l + 1];
    for (int l = 0; l + 1 < m.n_layers; ++l) {
      off += (int64_t)m.dims[l] * m.dims[l + 1];
      boff += m.dims[l + 1];
    }
    w = off;
    b = boff;
  };
  last(a.opacity, p.w_o2, p.b_o2);
  last(a.color, p.w_c2, p.b_c2);
  p.ldc2 = a.color.dims[a.color.n_layers];
  if (a.color_chn > 4) {  // the colour output layer on the matrix cores: its color_chn real columns, row stride = padded width
    p.co.w = p.w_c2;
    p.co.b = p.b_c2;
    p.co.rows_in = p.hc_w;
    p.co.cols = a.color_chn;
    p.co.ld = p.ldc2;
    p.co.ob = 1;
    p.co.bias = f;
    f += 32 * NB;
    p.co.img = img;
    img += loop_layer_bytes(p.hc_w, a.color_chn);
  }
  p.wo2 = f; f += 32 * NB;
  p.wc2 = f; f += 32 * NB * 4;
  p.hb = f; f += 8;
  p.inf = f; f += LOOP_N_INF;
  const int small_bytes = f * 4;  // multiple of 16
  for (int l = 0; l < p.n_t; ++l) p.t[l].img += small_bytes;
  for (int l = 0; l < p.n_o; ++l) p.o[l].img += small_bytes;
  for (int l = 0; l < p.n_c; ++l) p.c[l].img += small_bytes;
  p.co.img += small_bytes;
  p.img_end = small_bytes + img;
  static const int dbg = getenv("LP_MFMA_DEBUG") ? atoi(getenv("LP_MFMA_DEBUG")) : 0;
  p.dbg = dbg;
  p.seg_blocks = 1;
  p.seg_fwd = 0;
  p.relu_dump = g_relu_dump;  // test hook (NULL in every product call)
  p.dump_words = NB * ((tg ? 2 : p.n_t) + p.n_o + p.n_c) + 1;
  // the two-block backward's third tile (loop_layer_bwd): when the images leave the room  (LP_LOOP_NO_ZTILE: A/B)
  static const bool no_z = getenv("LP_LOOP_NO_ZTILE") != nullptr;
  p.tile_stride = LoopTile::PER_WAVE;
  p.z_delta = 0;
  if (NB == 2 && !no_z && (size_t)p.img_end + (size_t)WAVES * (LoopTile::PER_WAVE + LoopTile::Z_EXTRA) * 4 <= 160 * 1024) {
    p.tile_stride = LoopTile::PER_WAVE + LoopTile::Z_EXTRA;
    p.z_delta = (LoopTile::ZT - LoopTile::YT) * 4;
  }
  return p;
}

// words per (ray, sample) of the ReLU dump (include/lightplane_hip.h, lp_renderer_relu_dump_words)
int renderer_loop_dump_words(const LpRendererArgs& a) { return loop_params(a).dump_words; }

// Segment-parallel march of a small batch (same rules as renderer_mfma_segments, lp_renderer_mfma.hip): the looped kernels
// run one four-wave workgroup per CU, so 256 workgroups = 32 768 rays fill the chip once; below that a workgroup per
// (128 rays, segment) fills it.  Not with 5..32 colour channels (the state records hold four colour sums).
// (A shallow decoder's backward -- lp_renderer_loop_shallow.hip -- runs two workgroups per CU like the tuned family's: 512
// workgroups = 65 536 rays fill the chip, and the tuned family's threshold of 32 768 rays applies.)
static bool loop_is_shallow(const LpRendererArgs& a) {
  static const bool no_shallow = getenv("LP_LOOP_NO_SHALLOW") != nullptr;
  const LoopParams p = loop_params(a);
  return loop_nb(p.hid, a.grid.channels) == 1 && a.color_chn <= 4 && p.n_t <= 2 && p.n_o <= 1 && p.n_c <= 1 && !no_shallow;
}
int renderer_loop_segments(const LpRendererArgs& a) {
  static const int forced = getenv("LP_SEGMENTS") ? atoi(getenv("LP_SEGMENTS")) : -1;
  if (forced == 0 || a.march.num_samples_inf != 0 || a.stop_neg_log_t > 0.0f || a.color_chn > 4) return 1;
  const int n_seg = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
  if (n_seg < 2) return 1;
  if (forced < 0 && a.rays.n_rays > (loop_is_shallow(a) ? 32768 : 24576)) return 1;
  return n_seg;
}

// LP_SEG_LEN-sample blocks per segment: seg_blocks_for() (lp_host.h) -- every workgroup stages up to 69 KB of limb images and flushes
// its dW once
static int loop_seg_blocks(const LpRendererArgs& a, unsigned ray_blocks, unsigned resident = 256u) {
  const int n_rec = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
  return seg_blocks_for(ray_blocks, n_rec, resident);
}

static size_t loop_lds_bytes(const LoopParams& p, bool backward) {
  return (size_t)p.img_end + (backward ? (size_t)WAVES * p.tile_stride * 4 : 0);
}

bool renderer_loop_fits(const LpRendererArgs& a) {
  return loop_lds_bytes(loop_params(a), true) <= 160 * 1024;
}

static bool loop_triplane(const LpRendererArgs& a) {
  static const bool generic_grids = getenv("LP_MFMA_GENERIC_GRIDS") != nullptr;  // debugging aid
  return !generic_grids && is_canonical_triplane(a.grid);
}

int renderer_forward_loop(const LpRendererArgs& a, hipStream_t stream) {
  unsigned nb = loop_blocks(a);
  if (nb == 0) return LP_OK;
  LoopParams p = loop_params(a);
  static const bool seg_fwd = getenv("LP_SEG_FWD") == nullptr || atoi(getenv("LP_SEG_FWD")) != 0;
  const bool segf = a.seg_prefix && seg_fwd && !a.seg_forward_off && a.color_chn <= 4;
  if (segf) {
    const int n_rec = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
    p.seg_blocks = loop_seg_blocks(a, nb, loop_nb(p.hid, a.grid.channels) == 1 ? 512u : 256u);  // (NB = 1 forward: two workgroups per CU)
    p.seg_fwd = 1;
    nb *= (unsigned)((n_rec + p.seg_blocks - 1) / p.seg_blocks);
  }
  const size_t lds = loop_lds_bytes(p, false);
  const bool tg = a.color_grid.n_grids > 0, wc = a.color_chn > 4, tri = loop_triplane(a);
  const int NB = loop_nb(p.hid, a.grid.channels);
  int rc = LP_OK;
#define LP_LOOP_FWD(CV, NBV, TGV, WCV) rc = launch_fwd_loop<CV, NBV, TGV, WCV>(a, p, nb, lds, tri, stream)
  if (a.grid.channels == 64) {
    LP_LOOP_FWD(64, 2, false, false);
  } else if (a.grid.channels == 16) {
    if (NB == 2 && tg) LP_LOOP_FWD(16, 2, true, false);
    else if (NB == 2) LP_LOOP_FWD(16, 2, false, false);
    else if (tg && wc) LP_LOOP_FWD(16, 1, true, true);
    else if (tg) LP_LOOP_FWD(16, 1, true, false);
    else if (wc) LP_LOOP_FWD(16, 1, false, true);
    else LP_LOOP_FWD(16, 1, false, false);
  } else {
    if (NB == 2 && tg) LP_LOOP_FWD(32, 2, true, false);
    else if (NB == 2) LP_LOOP_FWD(32, 2, false, false);
    else if (tg && wc) LP_LOOP_FWD(32, 1, true, true);
    else if (tg) LP_LOOP_FWD(32, 1, true, false);
    else if (wc) LP_LOOP_FWD(32, 1, false, true);
    else LP_LOOP_FWD(32, 1, false, false);
  }
#undef LP_LOOP_FWD
  if (rc) return rc;
  if (segf && (rc = renderer_forward_combine_launch(a, p.seg_blocks, stream))) return rc;
  return check_launch("renderer_fwd_loop");
}

int renderer_backward_loop(const LpRendererArgs& a, hipStream_t stream) {
  unsigned nb = loop_blocks(a);
  if (nb == 0) return LP_OK;
  LoopParams p = loop_params(a);
  if (a.seg_prefix && a.color_chn <= 4) {  // small batch: one workgroup per (128 rays, segment)
    const int n_rec = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
    p.seg_blocks = loop_seg_blocks(a, nb, loop_is_shallow(a) ? 512u : 256u);
    nb *= (unsigned)((n_rec + p.seg_blocks - 1) / p.seg_blocks);
  }
  const size_t lds = loop_lds_bytes(p, true);
  const bool tg = a.color_grid.n_grids > 0, wc = a.color_chn > 4, tri = loop_triplane(a);
  const int NB = loop_nb(p.hid, a.grid.channels);
  int rc = LP_OK;
  static const bool no_shallow = getenv("LP_LOOP_NO_SHALLOW") != nullptr;  // A/B: the deep one-wave-per-SIMD instantiations for every shape
  if (NB == 1 && !wc && p.n_t <= 2 && p.n_o <= 1 && p.n_c <= 1 && !no_shallow) {
    rc = p.relu_dump ? renderer_backward_loop_shallow_dump(a, p, nb, lds, tri, stream) : renderer_backward_loop_shallow(a, p, nb, lds, tri, stream);
    if (rc) return rc;
    return check_launch("renderer_bwd_loop (shallow)");
  }
  (void)tg;
  rc = p.relu_dump ? renderer_backward_loop_deep_dump(a, p, NB, nb, lds, tri, stream) : loop_bwd_table_deep<false>(a, p, NB, nb, lds, tri, stream);
  if (rc) return rc;
  return check_launch("renderer_bwd_loop");
}

// what this translation unit's backward computes in (lp_build_info)
const char* build_info_loop_deep() {
#define LP_STR2(x) #x
#define LP_STR(x) LP_STR2(x)
  return "{\"dx_limbs\": \"" LP_STR(LP_DX_LIMBS) " (two-block / <= 2 trunk + 1 hidden head layers), 3 (deeper chains)\", \"dw\": "
#if LP_LOOP_DW_BF16
         "\"two-limb bf16 operands, v_mfma_f32_16x16x32_bf16\"}";
#else
         "\"fp32 operands, v_mfma_f32_16x16x4_f32\"}";
#endif
}

}  // namespace lp
