// Device-resident corpus for item2vec: what memory.Corpus + dictionary.Dictionary hold in the
// reference (corpus/memory/memory.go:25-102, corpus/dictionary/dictionary.go:21-81), for integer tokens.
#pragma once
#include <cstdint>
#include <mutex>

#include "common.h"

struct goctr_corpus {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  int64_t capacity = 0;
  int64_t n_words = 0;      // Corpus.Len() = maxLen: every word read, filtered or not
  int64_t V = 0;            // Dictionary.Len()
  int64_t n_indexed = 0;    // len(IndexedDoc())
  bool built = false;
  goctr::DevBuf<long long> keys;      // [capacity] tokens in stream order
  goctr::DevBuf<int> idoc;            // [n_words] dictionary id of every word (memory.go:85-88)
  goctr::DevBuf<int> indexed;         // [n_indexed] idoc minus the MinCount / MaxCount drops (memory.go:53-62)
  goctr::DevBuf<long long> id2key;    // [V] Dictionary.id2word
  goctr::DevBuf<long long> cfs;       // [V] Dictionary.cfs
  std::mutex mu;
};
