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
    return y.size() == 118 ? 0 : 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "goctr: %s\n", e.what());
    return 1;
  }
}
