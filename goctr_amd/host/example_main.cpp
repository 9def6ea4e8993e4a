// example_main.cpp -- the reference's model_test.go:TestMultiModel flow (DIN train -> predict 118 rows @ 20)
// written against the C++ host mirror.  Built by the CPU test suite (link check) and run by a GPU test.
#include <cstdio>

#include "goctr.hpp"

int main() {
  using namespace goctr;
  const int U = 5, T = 3, D = 7, C = 5, N = 2000, XC = U + T * D + D + C;
  std::mt19937 g(42);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  std::vector<float> X((size_t)N * XC), Y((size_t)N);
  for (auto& v : X) v = ud(g);
  for (int i = 0; i < N; ++i) Y[i] = X[(size_t)i * XC] + X[(size_t)i * XC + XC - 1] > 1.f ? 1.f : 0.f;
  try {
    auto si = recommend::SampleInfo::FromDims(U, T, D, C);
    din::DinNet net(U, T, D, D, C);
    net.InitGaussian(7);
    auto costs = model::Train(U, T, D, D, C, N, 200, 5, 0, si, X.data(), XC, Y.data(), net);
    auto y = model::Predict(net, 118, 20, si, X.data(), XC);
    std::printf("epochs %zu last cost %.6f pred[0] %.6f n %zu\n", costs.size(), costs.back(), y[0], y.size());
    // recommend.Rank (rcmd.go:248-275) over resident tables: 3 users with 4-item histories, 6 items
    std::vector<int64_t> off{0, 4, 8, 12}, ts{40, 30, 20, 10, 40, 30, 20, 10, 40, 30, 20, 10};
    std::vector<int32_t> hist{0, 1, 2, 3, 1, 2, 3, 4, 2, 3, 4, 5};
    std::vector<float> ut(3 * U), it(6 * C), emb(6 * D);
    for (auto& v : ut) v = ud(g);
    for (auto& v : it) v = ud(g);
    for (auto& v : emb) v = ud(g) - 0.5f;
    recommend::RecSys rs(off, hist, ts, ut, U, it, C, emb, D);
    auto scores = recommend::Rank(net, rs, 1, {0, 3, 99, 5}, 35);      // item 99 has no features: zero row
    std::printf("rank n %zu s0 %.6f s2 %.6f\n", scores.size(), scores[0].Score, scores[2].Score);
    return y.size() == 118 && scores.size() == 4 ? 0 : 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "goctr: %s\n", e.what());
    return 1;
  }
}
