// Layout of the drift table (xdfm_drift_tab_floats / xdfm_drift_tab_init / xdfm_opt_advance), shared by
// the kernels that build it (optim_loss_kernels.hip) and the ones that read it (embed_kernels.hip).
//
// TF's dense Adam moves a row that has no token in a batch all the same (base_model.py:290, :314-322 with
// the dense table gradient of SURVEY.md 8a A2):  m <- b1 m, v <- b2 v, p <- p - alpha_j m / (sqrt(v) + eps).
// Over G absent steps after step s, with g = sqrt(b2):
//     p_G = p - m * sum_{i=1..G} c_i / (sqrt(v) + eps g^-i),      c_i = alpha_{s+i} (b1 / g)^i
// eps g^-i moves by about 1 % over the steps that carry weight, so 1 / (sqrt(v) + e + d_i) is expanded to
// second order in d_i = eps (g^-i - g^-kDriftC) around e = eps g^-kDriftC;  with y = 1 / (sqrt(v) + e)
//     sum = y C0 - y^2 C1 + y^3 C2,      C_k(s, G) = sum_i c_i d_i^k
// (relative error of the sum below 1e-6: tools/drift_sim.py).  The table holds (C0, C1, C2, 0) as float4:
//     cur[2][kDriftG]   G < kDriftG steps missed as of optimizer step t (s = t - G), in slot t & 1; every
//                       xdfm_opt_advance forms the new t's slot from the other one (entry G from entry G - 1 plus
//                       step t's term, with the alpha_t the update itself uses), so readers of step t's slot
//                       may still run while t + 1's is written
//     hist[kDriftS + 1] G >= kDriftG (the terms beyond 255 steps weigh 0.9^255): a function of s alone, built
//                       once; entry kDriftS serves every s >= kDriftS, where alpha_j = lr to the last bit
#pragma once

namespace xdfm {
constexpr int kDriftG = 256;
constexpr int kDriftS = 16384;
constexpr int kDriftC = 8;
constexpr int kDriftFloat4 = 2 * kDriftG + kDriftS + 1;
}  // namespace xdfm
