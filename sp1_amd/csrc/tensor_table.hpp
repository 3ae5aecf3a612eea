// sp1_amd/csrc/tensor_table.hpp — a `Message<Tensor>` (several column-major tensors of equal height)
// flattened into one device table of column base pointers, so kernels index "column g of the
// concatenated row" with one wave-uniform (scalar-cache) pointer load.
#pragma once
#include <vector>

#include "common.hpp"

namespace sp1hip {

constexpr int MAX_TENSORS = 256;      // (the table travels as a kernel argument: 12 bytes per tensor of the 4 KB a launch may pass)

struct TensorTable {
    const uint32_t* base[MAX_TENSORS];
    uint32_t col_start[MAX_TENSORS + 1];  // prefix sums of widths
    int n;
};

// Validates the message and fills `tab`; returns total width through *total_width.
int make_tensor_table(const sp1hip_tensor_t* tensors, int n_tensors, TensorTable* tab, uint32_t* total_width);

// Enqueues the expansion of `tab` into d_cols[total_width] (device pointers to each column).
int expand_columns_async(const TensorTable& tab, uint32_t total_width, uint64_t height, const uint32_t** d_cols,
                         hipStream_t stream);

// One launch of the split leaf hash (merkle.hip): columns [c0, c0 + width) of the concatenated row, ready
// once tensors 0 .. last_tensor are encoded.
struct LeafPart { int last_tensor; uint32_t c0, width; };

// RAII scratch from the stream-keyed arena (the block goes back to the free list when the launch
// has been enqueued; the next user on the same stream is ordered behind it).
struct AsyncScratch {
    void* p = nullptr;
    hipStream_t s = nullptr;
    size_t n = 0;
    int alloc(size_t bytes, hipStream_t stream) {
        s = stream;
        n = bytes;
        return arena_alloc(&p, bytes, stream);
    }
    ~AsyncScratch() { arena_free(p, n, s); }
};

}  // namespace sp1hip
