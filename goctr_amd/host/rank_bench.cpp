// rank_bench.cpp -- recommend.Rank as the reference's HTTP API calls it (recommend/api.go:106-131 -> rcmd.go:248-275):
// one user and a short itemIdList per request, from concurrent handler goroutines.  Here: host threads (std::thread
// stands in for the goroutines of a cgo host) each calling goctr_rank in a closed loop on ONE model and ONE recsys.
//
//   rank_bench [--threads 1,8] [--n 32,256,2048] [--seconds 0.4] [--kind din|youtube] [--coalesce 1|0|both]
//
// Prints ONE JSON object: per (n, threads, coalesce) the request rate, rows/s and the latency distribution, plus a
// bit-equality check of every answer against the single-threaded answer for the same request (coalescing must not
// change a score).  bench.py runs it on rank 0 and merges the object into its line (rank_* fields).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <sstream>
#include <thread>

#include "goctr.hpp"

using clk = std::chrono::steady_clock;

static std::vector<int> parse_list(const char* s) {
  std::vector<int> v;
  std::stringstream ss(s);
  std::string tok;
  while (std::getline(ss, tok, ',')) if (!tok.empty()) v.push_back(std::atoi(tok.c_str()));
  return v;
}

int main(int argc, char** argv) {
  using namespace goctr;
  std::vector<int> threads{1, 8}, ns{32, 256, 2048};
  double seconds = 0.4;
  std::string kind = "din", coalesce = "both";
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    if (k == "--threads") threads = parse_list(argv[i + 1]);
    else if (k == "--n") ns = parse_list(argv[i + 1]);
    else if (k == "--seconds") seconds = std::atof(argv[i + 1]);
    else if (k == "--kind") kind = argv[i + 1];
    else if (k == "--coalesce") coalesce = argv[i + 1];
  }
  try {
    // BASELINE configs[2] shapes (cfg3): U 52, T 50, D 16, C 53, vocab 26 744; 8192 users with 20..120 behaviours
    const bool yt = kind == "youtube";
    const int U = 52, T = 50, D = yt ? 64 : 16, C = 53;
    const int64_t V = yt ? 1000000 : 26744, n_users = 8192;
    std::mt19937_64 g(7);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    std::vector<int64_t> off(n_users + 1, 0), ts;
    std::vector<int32_t> hist;
    for (int64_t u = 0; u < n_users; ++u) {
      const int len = 20 + (int)(g() % 101);
      off[u + 1] = off[u] + len;
      int64_t t = (int64_t)1 << 30;
      for (int k = 0; k < len; ++k) { t -= 1 + (int64_t)(g() % 1000); ts.push_back(t); hist.push_back((int32_t)(g() % V)); }
    }
    std::vector<float> ut((size_t)n_users * U), it((size_t)V * C), emb((size_t)V * D);
    for (auto& v : ut) v = ud(g);
    for (auto& v : it) v = ud(g);
    for (auto& v : emb) v = (ud(g) - 0.5f) * 0.5f;
    recommend::RecSys rs(off, hist, ts, ut, U, it, C, emb, D);
    model::CtrNet net(yt ? GOCTR_YOUTUBE : GOCTR_DIN, U, T, D, D, C);
    net.InitGaussian(11);

    const int64_t now = ((int64_t)1 << 30) - 50000;
    std::printf("{\"workload\": \"goctr_rank: one user x n candidate items per call (recommend/api.go:106-131), %s cfg shapes "
                "(U %d, T %d, D %d, C %d, vocab %lld), closed loop of host threads on one model\", \"results\": [",
                kind.c_str(), U, T, D, C, (long long)V);
    bool first = true, all_equal = true;
    std::vector<std::string> modes;
    if (coalesce == "both") modes = {"1", "0"}; else modes = {coalesce};
    for (const std::string& mode : modes) {
      setenv("GOCTR_SERVE_COALESCE", mode == "0" ? "0" : "1024", 1);
      for (int n : ns) {
        // the requests: 64 (user, candidate list) pairs, with their single-threaded answers
        const int NREQ = 64;
        std::vector<int32_t> users(NREQ);
        std::vector<std::vector<int32_t>> items(NREQ, std::vector<int32_t>((size_t)n));
        std::vector<std::vector<float>> want(NREQ, std::vector<float>((size_t)n));
        for (int q = 0; q < NREQ; ++q) {
          users[q] = (int32_t)(g() % n_users);
          for (auto& v : items[q]) v = (int32_t)(g() % V);
          check(goctr_rank(net.Vm(), rs.handle(), users[q], items[q].data(), n, now, 4096, want[q].data(), nullptr, nullptr));
        }
        for (int nt : threads) {
          std::atomic<bool> stop{false};
          std::atomic<int> bad{0};
          std::vector<std::vector<float>> lat((size_t)nt);
          std::vector<std::thread> th;
          const auto t0 = clk::now();
          for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
              std::vector<float> y((size_t)n);
              int q = t * 7;
              while (!stop.load(std::memory_order_relaxed)) {
                q = (q + 1) % NREQ;
                const auto a = clk::now();
                if (goctr_rank(net.Vm(), rs.handle(), users[q], items[q].data(), n, now, 4096, y.data(), nullptr, nullptr)) { bad += 1000000; break; }
                const auto b = clk::now();
                lat[t].push_back(std::chrono::duration<float, std::micro>(b - a).count());
                if (std::memcmp(y.data(), want[q].data(), sizeof(float) * (size_t)n) != 0) bad++;
              }
            });
          std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
          stop = true;
          for (auto& x : th) x.join();
          const double dt = std::chrono::duration<double>(clk::now() - t0).count();
          std::vector<float> all;
          for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
          std::sort(all.begin(), all.end());
          const size_t k = all.size();
          double mean = 0;
          for (float v : all) mean += v;
          mean /= std::max<size_t>(k, 1);
          if (bad.load()) all_equal = false;
          std::printf("%s{\"n\": %d, \"threads\": %d, \"coalesce\": %s, \"calls\": %zu, \"rank_qps\": %.1f, \"rows_per_s\": %.1f, "
                      "\"latency_us\": {\"mean\": %.1f, \"p50\": %.1f, \"p90\": %.1f, \"p99\": %.1f, \"p999\": %.1f, \"max\": %.1f, \"min\": %.1f}, "
                      "\"mismatched_calls\": %d}",
                      first ? "" : ", ", n, nt, mode == "0" ? "false" : "true", k, k / dt, k * (double)n / dt, mean,
                      k ? all[k / 2] : 0.f, k ? all[std::min(k - 1, (size_t)(k * 0.90))] : 0.f, k ? all[std::min(k - 1, (size_t)(k * 0.99))] : 0.f,
                      k ? all[std::min(k - 1, (size_t)(k * 0.999))] : 0.f, k ? all[k - 1] : 0.f, k ? all[0] : 0.f, bad.load());
          first = false;
        }
      }
    }
    std::printf("], \"bit_equal_to_single_threaded\": %s}\n", all_equal ? "true" : "false");
    return all_equal ? 0 : 3;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rank_bench: %s\n", e.what());
    return 1;
  }
}
