// knn_bench.cpp -- Searcher.Search (feature/embedding/search/search.go:92-134) as a COMPILED host drives it: a closed loop of
// goctr_searcher_search calls over one resident catalogue, no interpreter between the calls (bench.py's own loop pays ~10 us of
// CPython per call -- numpy allocations, ctypes marshalling -- on top of a ~40 us call; a Go host does not).
//
//   knn_bench [--items 1000000] [--dim 16] [--queries 64] [--k 10] [--steps 200] [--warmup 20] [--regions 9] [--seed 42]
//
// Items and queries are N(0,1) (std::mt19937_64 + std::normal_distribution); prints ONE JSON object: the median region, every
// region, queries/s, and a checksum of the last call's neighbour lists (two runs with the same seed must print the same one).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/goctr.h"

using clk = std::chrono::steady_clock;

int main(int argc, char** argv) {
  long long V = 1000000; int D = 16, Q = 64, k = 10, steps = 200, warm = 20, regions = 9; unsigned long long seed = 42;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string a = argv[i];
    if (a == "--items") V = std::atoll(argv[i + 1]);
    else if (a == "--dim") D = std::atoi(argv[i + 1]);
    else if (a == "--queries") Q = std::atoi(argv[i + 1]);
    else if (a == "--k") k = std::atoi(argv[i + 1]);
    else if (a == "--steps") steps = std::atoi(argv[i + 1]);
    else if (a == "--warmup") warm = std::atoi(argv[i + 1]);
    else if (a == "--regions") regions = std::atoi(argv[i + 1]);
    else if (a == "--seed") seed = std::strtoull(argv[i + 1], nullptr, 10);
  }
  auto die = [](const char* what) { std::fprintf(stderr, "knn_bench: %s: %s\n", what, goctr_last_error()); std::exit(1); };
  if (goctr_init(0)) die("goctr_init");
  std::mt19937_64 g(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> items((size_t)V * D), queries((size_t)Q * D);
  for (auto& x : items) x = nd(g);
  for (auto& x : queries) x = nd(g);
  goctr_searcher* s = nullptr;
  if (goctr_searcher_create(items.data(), V, D, &s)) die("goctr_searcher_create");
  std::vector<int64_t> idx((size_t)Q * k);
  std::vector<double> sim((size_t)Q * k);
  std::vector<int> cnt((size_t)Q);
  auto call = [&]() {
    if (goctr_searcher_search(s, queries.data(), Q, k, nullptr, idx.data(), sim.data(), cnt.data())) die("goctr_searcher_search");
  };
  for (int i = 0; i < warm; ++i) call();
  goctr_sync();
  std::vector<double> reg;
  for (int r = 0; r < regions; ++r) {
    const auto t0 = clk::now();
    for (int i = 0; i < steps; ++i) call();
    goctr_sync();
    reg.push_back(std::chrono::duration<double, std::milli>(clk::now() - t0).count());
  }
  std::vector<double> sorted = reg;
  std::sort(sorted.begin(), sorted.end());
  const double med = sorted[(sorted.size() - 1) / 2];
  unsigned long long sum = 1469598103934665603ULL;                 // FNV-1a over the neighbour indices and counts
  for (auto v : idx) { sum ^= (unsigned long long)v; sum *= 1099511628211ULL; }
  for (auto v : cnt) { sum ^= (unsigned long long)v; sum *= 1099511628211ULL; }
  std::printf("{\"items\": %lld, \"dim\": %d, \"queries_per_call\": %d, \"k\": %d, \"steps\": %d, \"warmup\": %d, \"us_per_call\": %.3f, "
              "\"queries_per_s\": %.1f, \"timed_regions_ms\": [", V, D, Q, k, steps, warm, med * 1e3 / steps, (double)steps * Q / (med * 1e-3));
  for (size_t i = 0; i < reg.size(); ++i) std::printf("%s%.4f", i ? ", " : "", reg[i]);
  std::printf("], \"neighbours_fnv1a\": \"%016llx\"}\n", sum);
  goctr_searcher_destroy(s);
  return 0;
}
