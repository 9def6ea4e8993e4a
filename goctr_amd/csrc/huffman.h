// huffman.h -- interface of huffman.hip (the Huffman build with the device: rocPRIM sort + host merge + device path fill).
#pragma once
#include <cstdint>

#include "common.h"

namespace goctr {

// counts_host [V] >= 0.  Leaves off [V + 1], nodes / codes [total] resident on the calling thread's engine.  parts_ms (may be
// null) = {sort + copy of the sorted counts to the host, host merge, path lengths + prefix sum + fill, total}.
int huffman_build_device(const long long* counts_host, int64_t V, int max_depth, DevBuf<long long>& off, DevBuf<int>& nodes,
                         DevBuf<unsigned char>& codes, long long* total_out, double parts_ms[4]);

}  // namespace goctr
