// The host-side matrix handle of the C ABI's ingestion entry points (scanrs_h5_*, scanrs_mtx_read): what the reference
// calls GenericFeatureBarcodeMatrix (scan-types/src/matrix.rs:8-15) with the matrix as plain CSR / CSC arrays.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "scanrs_amd.h"

struct scanrs_h5_matrix {
    std::string name;
    std::vector<std::string> barcodes, feature_ids, feature_names, feature_types;
    bool has_matrix = false;
    int storage = SCANRS_CSC;
    uint64_t rows = 0, cols = 0, nnz = 0;
    std::vector<uint64_t> indptr;
    std::vector<uint32_t> indices, values;
    std::vector<uint64_t> removed;
};
