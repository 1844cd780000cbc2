/* Binding liblightplane_hip.so from plain C: the drop-in boundary of include/lightplane_hip.h without Python.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_example.c -o c_abi_example -ldl
 *   ./c_abi_example lightplane_amd/liblightplane_hip.so
 *
 * The program fills an LpRendererArgs for the benchmark decoder (triplane 64^2 x 16 ch, MLPs 2 x 32, RGB), asks the
 * library which kernel family it would run and validates the arguments with a zero-ray launch (nothing is
 * launched, so this part also runs on a machine without a GPU).  A real caller passes device pointers obtained
 * from hipMalloc / its framework's tensors, a hipStream_t, and zero-filled gradient buffers for the backward
 * (INTEGRATION.md). */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "lightplane_hip.h"

typedef int (*version_fn)(void);
typedef const char* (*error_fn)(void);
typedef int (*sizeof_fn)(int);
typedef int (*renderer_fn)(const LpRendererArgs*, void*);
typedef int (*family_fn)(const LpRendererArgs*);

static LpMlp mlp(int n_layers, int d0, int d1, int d2, int64_t offset) {
  LpMlp m;
  memset(&m, 0, sizeof m);
  m.n_layers = n_layers;
  m.dims[0] = d0; m.dims[1] = d1; m.dims[2] = d2;
  m.offset = offset;
  return m;
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "lightplane_amd/liblightplane_hip.so";
  void* lib = dlopen(path, RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  version_fn lp_version_ = (version_fn)dlsym(lib, "lp_version");
  error_fn lp_last_error_ = (error_fn)dlsym(lib, "lp_last_error");
  sizeof_fn lp_abi_sizeof_ = (sizeof_fn)dlsym(lib, "lp_abi_sizeof");
  renderer_fn lp_renderer_forward_ = (renderer_fn)dlsym(lib, "lp_renderer_forward");
  family_fn lp_renderer_kernel_family_ = (family_fn)dlsym(lib, "lp_renderer_kernel_family");
  if (!lp_version_ || !lp_last_error_ || !lp_abi_sizeof_ || !lp_renderer_forward_ || !lp_renderer_kernel_family_) {
    fprintf(stderr, "missing symbol\n");
    return 2;
  }
  if (lp_abi_sizeof_(5) != (int)sizeof(LpRendererArgs)) { fprintf(stderr, "header / library mismatch\n"); return 2; }

  LpRendererArgs a;
  memset(&a, 0, sizeof a);
  const int C = 16, H = 32, R = 64;
  a.grid.n_grids = 3;
  a.grid.channels = C;
  a.grid.n_rows = 3 * (int64_t)R * R;
  a.grid.grids[0].B = 1; a.grid.grids[0].D = 1; a.grid.grids[0].H = R; a.grid.grids[0].W = R; a.grid.grids[0].row_offset = 0;
  a.grid.grids[1].B = 1; a.grid.grids[1].D = R; a.grid.grids[1].H = 1; a.grid.grids[1].W = R; a.grid.grids[1].row_offset = (int64_t)R * R;
  a.grid.grids[2].B = 1; a.grid.grids[2].D = R; a.grid.grids[2].H = R; a.grid.grids[2].W = 1; a.grid.grids[2].row_offset = 2 * (int64_t)R * R;
  a.march.num_samples = 128;
  a.march.disparity_at_inf = 1e-5;
  /* flat parameter vector: [trunk W0 W1 b0 b1][opacity W0 W1 b0 b1][colour W0 W1 b0 b1], colour padded to 16 columns */
  const int64_t n_trunk = (int64_t)C * H + H * H + 2 * H, n_op = (int64_t)H * H + H + H + 1, n_col = (int64_t)H * H + H * 16 + H + 16;
  a.trunk = mlp(2, C, H, H, 0);
  a.opacity = mlp(2, H, H, 1, n_trunk);
  a.color = mlp(2, H, H, 16, n_trunk + n_op);
  a.n_mlp_params = n_trunk + n_op + n_col;
  a.mlp_params = (const float*)0x1000; /* never dereferenced with zero rays */
  a.color_chn = 3;
  a.gain = 1.0f;
  a.rays.encoding_dim = H;
  a.rays.n_rays = 0;

  const int family = lp_renderer_kernel_family_(&a);
  const int rc = lp_renderer_forward_(&a, NULL);
  printf("liblightplane_hip %d.%d.%d: kernel family %d (1 = MFMA width 32), zero-ray launch rc = %d\n", lp_version_() / 1000,
         lp_version_() / 100 % 10, lp_version_() % 100, family, rc);
  a.march.num_samples = 0; /* an invalid argument comes back as a code + message, never an abort */
  const int bad = lp_renderer_forward_(&a, NULL);
  printf("invalid argument: rc = %d (%s)\n", bad, lp_last_error_());
  dlclose(lib);
  return (family == 1 && rc == 0 && bad == LP_EINVAL) ? 0 : 1;
}
