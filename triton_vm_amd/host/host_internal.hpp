// host_internal.hpp -- helpers shared by the translation units of the C++ host (triton_host.cpp, sharded_host.cpp).
// Not part of the interface: triton_host.hpp is.
#pragma once
#include <string>
#include <vector>

#include "triton_host.hpp"

namespace triton_vm {

static const u64 NUM_MAIN = TVM_NUM_MAIN_COLUMNS, NUM_AUX = TVM_NUM_AUX_COLUMNS, NUM_SAMPLED_CHALLENGES = TVM_NUM_CHALLENGES - 4;

u64 bfe_add(u64 a, u64 b);
Xfe xfe_add(const Xfe& a, const Xfe& b);
Xfe xfe_mul(const Xfe& a, const Xfe& b);
Xfe xfe_scale(const Xfe& a, u64 s);
std::vector<Xfe> xfe_powers(const Xfe& x, u64 first, u64 n);
unsigned bit_length(u64 v);
std::vector<u64> merkle_root(const Context& c, const DeviceBuffer& nodes);  // node 1; drains the stream
// [twenty-first MerkleTree::authentication_structure, restated]: heap indices of the nodes a verifier cannot compute itself
std::vector<u64> auth_node_indices(u64 n_leaves, const std::vector<u64>& indices);
std::vector<Xfe> derive_challenges(std::vector<Xfe> sampled, const Claim& claim);  // Challenges::new (challenges.rs:85-121)
std::vector<u64> trace_randomizers_host(const uint8_t table_seed[32], u64 n_cols, u64 h, int fk);
DeviceBuffer upload(const Context& c, const std::vector<u64>& host);

// Many gathers with one round trip (tvm_gather_elements_batch): jobs are queued with their index lists, run() fills `out`.
struct GatherBatch {
    struct Job {
        const u64* src;
        uint32_t words;
        std::vector<u64> idx, out;
    };
    std::vector<Job> jobs;
    size_t add(const u64* src, uint32_t words, std::vector<u64> idx) {
        jobs.push_back(Job{src, words, std::move(idx), {}});
        return jobs.size() - 1;
    }
    void run(const Context& c) {
        std::vector<const uint64_t*> src, idx;
        std::vector<uint32_t> words;
        std::vector<uint64_t> n;
        std::vector<uint64_t*> out;
        for (Job& j : jobs) {
            j.out.assign(j.idx.size() * j.words, 0);
            src.push_back(j.src);
            words.push_back(j.words);
            idx.push_back(j.idx.data());
            n.push_back(j.idx.size());
            out.push_back(j.out.data());
        }
        c.check(tvm_gather_elements_batch(c.raw(), (uint32_t)jobs.size(), src.data(), words.data(), idx.data(), n.data(), out.data()),
                "tvm_gather_elements_batch");
    }
};


}  // namespace triton_vm
