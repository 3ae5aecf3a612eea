// sp1_amd/csrc/stacked_data.hpp — the prover data behind sp1hip_stacked_data_t, shared by stacked.hip
// (commit) and jagged.hip (evaluation proof): `StackedBasefoldProverData`
// (/root/reference/slop/crates/stacked/src/prover.rs:L20-L31) plus, when it came from
// sp1hip_jagged_commit, the `JaggedProverData` fields (/root/reference/slop/crates/jagged/src/prover.rs:L36-L46).
#pragma once
#include <atomic>
#include <vector>

#include "common.hpp"

struct sp1hip_stacked_data_s {
    void* d_dense = nullptr;             // dense column-major concatenation of the tables, zero-padded
    hipStream_t stream = nullptr;
    sp1hip_basefold_data_t* basefold = nullptr;
    std::vector<sp1hip_tensor_t> batches;    // slices of d_dense: [2^lsh x w] stacked batches
    uint64_t area = 0, padded = 0;
    int log_stacking_height = 0;
    uint32_t commit[8];                  // stacked (inner) commitment = JaggedProverData.original_commitment
    // jagged wrapper
    bool jagged = false;
    int max_log_row_count = 0;
    std::vector<uint64_t> row_counts, column_counts;   // per table, the two padding tables appended
    uint64_t padding_column_count = 0;
    uint32_t jagged_commit[8];
    std::atomic<bool> foreign_use{false};            // read on a stream other than `stream` (see sp1hip_basefold_data_s)
    ~sp1hip_stacked_data_s() {
        if (foreign_use) (void)hipDeviceSynchronize();
        if (basefold) sp1hip_basefold_data_free(basefold);
        sp1hip::arena_free(d_dense, padded * 4, stream);
    }
};
