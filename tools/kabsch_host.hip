// Test harness (not part of the library): the alignment solver of csrc/cvf_kabsch.hpp compiled in hipcc's HOST pass, so that
// tests/test_kabsch_host.py can run the very source the kernels inline on the CPU against an fp64 SVD (no GPU needed).
//   hipcc -O2 -std=c++17 -shared -fPIC --offload-arch=gfx950 -Iinclude -Icolvars-finder_amd/csrc tools/kabsch_host.hip -o <out>.so
#include "cvf_kabsch.hpp"

void cvf_set_error(const char*, ...) {}
int cvf_check_launch(const char*) { return 0; }

// H [n][9] row-major fp64 -> R [n][9], Kinv [n][6] (fp32, as the kernels store them), R0 [n][9] (the fp32 stage alone) and
// resolved [n] (0: the fp32 stage handed the frame to the fp64 stage)
extern "C" void cvf_test_kabsch(const double* H, int n, float* R, float* Kinv, float* R0, int* resolved) {
  for (int b = 0; b < n; ++b) {
    double Hm[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Hm[i][j] = H[9 * b + 3 * i + j];
    KabschOut ko;
    kabsch_from_H(Hm, ko);
    for (int i = 0; i < 9; ++i) R[9 * b + i] = ko.R[i];
    for (int i = 0; i < 6; ++i) Kinv[6 * b + i] = ko.Kinv[i];
    float g[3][3];
    resolved[b] = kabsch_guess<float, 3>(Hm, g) ? 1 : 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R0[9 * b + 3 * i + j] = g[i][j];
  }
}

// the rotation-only variant (features without the derivative's K^-1: one Newton step)
extern "C" void cvf_test_kabsch_rotation_only(const double* H, int n, float* R) {
  for (int b = 0; b < n; ++b) {
    double Hm[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Hm[i][j] = H[9 * b + 3 * i + j];
    KabschOut ko;
    kabsch_from_H<false>(Hm, ko);
    for (int i = 0; i < 9; ++i) R[9 * b + i] = ko.R[i];
  }
}
