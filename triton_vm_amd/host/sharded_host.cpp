// sharded_host.cpp -- Prover::prove with the extended master tables split by cosets of the trace domain, over the GPUs of a
// node (one rank per GPU, collectives through a tvmh_comm) and / or coset by coset on one GPU.  See triton_host.hpp.
//
// The decomposition is the reference's own: compute_quotient_segments_with_jit_lde (stark.rs:805-1006) evaluates the tables on
// one coset of the trace domain at a time, the JIT branches of hash_all_ldt_domain_rows (master_table.rs:470-503) and
// reveal_rows (:556-609) do the same for hashing and opening, and the successor of a row stays in its coset
// (master_table.rs:1305-1306).  The extended rows i = g (mod G) of a domain are the domain (offset * generator^g,
// generator^G, length / G), so every per-group step is the ordinary C-ABI call on that domain; no kernel knows about ranks
// or passes.  Group g = rank + world * pass.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "host_internal.hpp"
#include "triton_host.hpp"

namespace triton_vm {

ArithmeticDomain coset_group(const ArithmeticDomain& d, u64 g, u64 G) {
    if (!G || d.length % G) throw Error(TVM_ERR_INVALID_ARGUMENT, "coset_group: the group count divides the domain length");
    return {mont_mul(d.offset, mont_pow(d.generator, g)), mont_pow(d.generator, G), d.length / G};
}

ShardedProver::ShardedProver(const Context& c, const StarkParameters& p, const tvmh_comm* comm, unsigned passes, const u64* d_main_trace,
                             const u64* d_main_randomizers, const u64* d_aux_trace, const u64* d_aux_randomizers,
                             const std::vector<Xfe>& quotient_randomizer, const Claim& claim)
    : c_(c), p_(p), comm_(comm), passes_(passes ? passes : 1), claim_(claim),
      main_(c, 1, d_main_trace, p.trace.length, NUM_MAIN, d_main_randomizers, p.h, p.trace, p.quotient, p.ldt),
      aux_(c, 3, d_aux_trace, p.trace.length, NUM_AUX, d_aux_randomizers, p.h, p.trace, p.quotient, p.ldt),
      quotient_randomizer_(quotient_randomizer) {
    if (quotient_randomizer.size() != p.num_quotient_randomizers) throw Error(TVM_ERR_INVALID_ARGUMENT, "quotient randomizer length");
    const u64 world = comm ? comm->world : 1, groups = world * passes_;
    const u64 x_ldt = p.ldt.length / p.trace.length, x_quot = p.quotient.length / p.trace.length;
    if (!world || (world & (world - 1)) || (passes_ & (passes_ - 1)) || x_ldt % groups || x_quot % groups || p.quotient.length > p.ldt.length)
        throw Error(TVM_ERR_INVALID_ARGUMENT, "coset sharding needs world size x passes (powers of two) to divide |LDT| / |trace| and |quotient| / |trace|");
    if (passes_ > 1 && p.quotient.length != p.ldt.length)
        throw Error(TVM_ERR_UNSUPPORTED, "coset-wise evaluation needs |quotient| == |LDT|");
    if (comm && comm->rank >= comm->world) throw Error(TVM_ERR_INVALID_ARGUMENT, "communicator rank");
}

typedef std::vector<u64> Words;

struct ShardedRun {
    struct Tree {  // a Merkle tree over n_leaves leaves: whole on this rank, or its lowest levels split over the ranks
        u64 n_leaves = 0;
        DeviceBuffer nodes;  // whole: [2 n][5] heap order; split: the subtree over this rank's contiguous leaf range, [2 n / R][5]
        bool split = false;
        Words top;           // split: the tree over the R subtree roots, [2 R][5], on every rank
        Words root() const { return Words(top.begin() + 5, top.begin() + 10); }
    };
    struct TableGuard {
        const Context& c;
        tvm_table* t = nullptr;
        ~TableGuard() { if (t) tvm_table_free(c.raw(), t); }
    };

    ShardedProver& sp;
    const Context& c;
    const StarkParameters& p;
    const tvmh_comm* comm;
    const u64 R, me, P;
    ProofStream ps;
    std::chrono::steady_clock::time_point t_stage;
    std::string stage;

    explicit ShardedRun(ShardedProver& s)
        : sp(s), c(s.c_), p(s.p_), comm(s.comm_), R(s.comm_ ? s.comm_->world : 1), me(s.comm_ ? s.comm_->rank : 0), P(s.passes_) {}

    // ---------------------------------------------------------------------------------------------- bookkeeping
    void close_stage() {
        if (!sp.profile || stage.empty()) return;
        (void)tvm_sync(c.raw());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_stage).count();
        for (auto& e : sp.stage_ms)
            if (e.first == stage) { e.second += ms; return; }
        sp.stage_ms.push_back({stage, ms});
    }
    void mark(const char* name) {
        close_stage();
        if (comm && comm->mark) comm->mark(comm->self, c.raw(), name);
        stage = name;
        t_stage = std::chrono::steady_clock::now();
    }
    void count(const char* what, u64 bytes) {
        for (auto& e : sp.exchanges)
            if (e.first == what) { e.second.first++; e.second.second += bytes; return; }
        sp.exchanges.push_back({what, {1, bytes}});
    }
    void comm_check(int32_t status, const char* what) const {
        if (status != TVM_OK) throw Error(status, std::string(what) + ": the communicator reported " + tvm_status_string(status));
    }
    void scatter(const u64* src, uint32_t w, u64 n, u64 stride, u64 offset, u64* dst) const {
        c.check(tvm_scatter_strided(c.raw(), src, w, n, stride, offset, dst), "tvm_scatter_strided");
    }

    // ---------------------------------------------------------------------------------------------- collectives
    // this rank's rows (global row i = local row i / R of rank i % R), n_local elements of w words -> all rows in row order
    DeviceBuffer gather_rows(DeviceBuffer&& local, u64 n_local, uint32_t w, const char* what) {
        if (!comm) return std::move(local);
        DeviceBuffer all(c, n_local * w * R), out(c, n_local * w * R);
        comm_check(comm->all_gather(comm->self, c.raw(), local.ptr(), all.ptr(), n_local * w), what);
        count(what, n_local * w * 8 * (R - 1));
        for (u64 r = 0; r < R; r++) scatter(all.ptr() + r * n_local * w, w, n_local, R, r, out.ptr());
        return out;
    }
    // one block of cap words per rank (zero-padded), host to host; the collective itself runs on device buffers
    std::vector<Words> all_gather_host(const Words& mine, u64 cap, const char* what) {
        if (mine.size() > cap) throw Error(TVM_ERR_INVALID_ARGUMENT, "all_gather_host: block longer than its capacity");
        if (!comm) return {mine};
        cap = std::max<u64>(cap, 1);
        Words block(cap, 0);
        std::copy(mine.begin(), mine.end(), block.begin());
        DeviceBuffer send(c, cap), recv(c, cap * R);
        c.check(tvm_memcpy_h2d(c.raw(), send.ptr(), block.data(), cap * 8), "tvm_memcpy_h2d");
        comm_check(comm->all_gather(comm->self, c.raw(), send.ptr(), recv.ptr(), cap), what);
        count(what, cap * 8 * (R - 1));
        const Words all = recv.download(0, cap * R);
        std::vector<Words> out(R);
        for (u64 r = 0; r < R; r++) out[r].assign(all.begin() + r * cap, all.begin() + (r + 1) * cap);
        return out;
    }
    // Several small gathers in ONE exchange (the openings of a proof: rows of three tables, leaves of the FRI rounds, the
    // authentication nodes of every tree).  A job names, per element, the rank that holds it (R = "known to every rank already")
    // and brings this rank's elements in order; run() moves every rank's blocks with one all-gather and fills `out` (w words per
    // element, in the order of `owner`; the elements with owner R are left for the caller).
    struct Exchange {
        struct Job {
            std::vector<u64> owner;
            uint32_t w;
            Words mine, out;
        };
        std::vector<Job> jobs;
        size_t add(std::vector<u64> owner, uint32_t w, Words mine) {
            jobs.push_back(Job{std::move(owner), w, std::move(mine), {}});
            return jobs.size() - 1;
        }
    };
    void run(Exchange& ex, const char* what) {
        std::vector<u64> offset(ex.jobs.size()), cap(ex.jobs.size());
        u64 block_words = 0;
        for (size_t j = 0; j < ex.jobs.size(); j++) {
            std::vector<u64> per_rank(R + 1, 0);
            for (u64 r : ex.jobs[j].owner) per_rank[r]++;
            cap[j] = *std::max_element(per_rank.begin(), per_rank.begin() + R);
            if (ex.jobs[j].mine.size() != per_rank[me] * ex.jobs[j].w) throw Error(TVM_ERR_INVALID_ARGUMENT, "exchange: a job's own elements");
            offset[j] = block_words;
            block_words += cap[j] * ex.jobs[j].w;
        }
        Words block(block_words, 0);
        for (size_t j = 0; j < ex.jobs.size(); j++) std::copy(ex.jobs[j].mine.begin(), ex.jobs[j].mine.end(), block.begin() + offset[j]);
        const std::vector<Words> all = all_gather_host(block, block_words, what);
        for (size_t j = 0; j < ex.jobs.size(); j++) {
            Exchange::Job& job = ex.jobs[j];
            job.out.assign(job.owner.size() * job.w, 0);
            std::vector<u64> taken(R, 0);
            for (size_t k = 0; k < job.owner.size(); k++) {
                const u64 r = job.owner[k];
                if (r < R) std::copy_n(all[r].begin() + offset[j] + taken[r]++ * job.w, job.w, job.out.begin() + k * job.w);
            }
        }
    }
    // elements (w words) at global indices of an array that is distributed by residue class: element i is local element
    // i / R of rank i % R; `fetch` reads this rank's elements at local indices.  -> the job (its `out` is in the order of `idx`)
    template <class Fetch>
    size_t add_distributed(Exchange& ex, const std::vector<u64>& idx, uint32_t w, Fetch fetch) {
        std::vector<u64> local, owner;
        for (u64 i : idx) {
            owner.push_back(i % R);
            if (i % R == me) local.push_back(i / R);
        }
        return ex.add(std::move(owner), w, local.empty() ? Words() : fetch(local));
    }

    // ---------------------------------------------------------------------------------------------- Merkle trees
    bool splits(u64 n_leaves) const { return comm && n_leaves >= std::max<u64>(sp.split_tree_min_leaves, 2 * R) && (n_leaves / R) % R == 0; }

    Tree whole_tree(const u64* d_leaves, u64 n, bool xfe_leaves) {
        Tree t;
        t.n_leaves = n;
        t.nodes = DeviceBuffer(c, 10 * n);
        if (xfe_leaves) c.check(tvm_codeword_merkle_tree(c.raw(), d_leaves, n, t.nodes.ptr()), "tvm_codeword_merkle_tree");
        else c.check(tvm_merkle_tree(c.raw(), d_leaves, n, t.nodes.ptr()), "tvm_merkle_tree");
        t.top.assign(10, 0);
        const Words root = merkle_root(c, t.nodes);
        std::copy(root.begin(), root.end(), t.top.begin() + 5);
        return t;
    }
    // leaves distributed by residue class (this rank: n / R of them; digests, or the XFE codeword elements that
    // Digest::from turns into leaves, fri.rs:343-347) -> the tree over all n leaves in row order
    Tree tree_from_local(DeviceBuffer&& local, u64 n, bool xfe_leaves, const char* what) {
        const uint32_t w = xfe_leaves ? 3 : 5;
        if (!splits(n)) {
            const DeviceBuffer full = gather_rows(std::move(local), n / R, w, what);
            return whole_tree(full.ptr(), n, xfe_leaves);
        }
        // The tree is built split: rank t needs the leaves of ITS contiguous range [t n/R, (t+1) n/R) only.  Of this rank's
        // rows (global row me + R a) those are the local rows a in [t n/R^2, (t+1) n/R^2): one all-to-all moves every
        // leaf once (1/R of what an all-gather moves), then the R received blocks are interleaved into row order.
        const u64 per = n / R, blk = per / R;
        DeviceBuffer recv(c, per * w), mine(c, per * w);
        comm_check(comm->all_to_all(comm->self, c.raw(), local.ptr(), recv.ptr(), blk * w), what);
        count(what, blk * w * 8 * (R - 1));
        for (u64 t = 0; t < R; t++) scatter(recv.ptr() + t * blk * w, w, blk, R, t, mine.ptr());
        Tree t;
        t.n_leaves = n;
        t.split = true;
        t.nodes = DeviceBuffer(c, 10 * per);
        if (xfe_leaves) c.check(tvm_codeword_merkle_tree(c.raw(), mine.ptr(), per, t.nodes.ptr()), "tvm_codeword_merkle_tree");
        else c.check(tvm_merkle_tree(c.raw(), mine.ptr(), per, t.nodes.ptr()), "tvm_merkle_tree");
        const std::vector<Words> roots = all_gather_host(merkle_root(c, t.nodes), 5, "subtree roots");
        t.top.assign(10 * R, 0);
        for (u64 r = 0; r < R; r++) std::copy_n(roots[r].begin(), 5, t.top.begin() + 5 * (R + r));
        for (u64 k = R - 1; k >= 1; k--) {  // hash_pair: the capacity of a fixed-length input is all ones (tip-0005.md:82)
            u64 state[16];
            std::copy_n(t.top.begin() + 10 * k, 10, state);
            for (int i = 10; i < 16; i++) state[i] = to_mont(1);
            tvm_host_tip5_permutation(state);
            std::copy_n(state, 5, t.top.begin() + 5 * k);
        }
        sp.split_trees_built++;
        return t;
    }
    // MerkleTree::authentication_structure for trees of the same shape opened at the same leaves (their node lists coincide): one
    // round trip to the device for all of them, and for split trees one job of the exchange (an element = the node of every tree).
    struct AuthJob {
        std::vector<const Tree*> trees;
        std::vector<u64> need;
        size_t job = 0;
        bool exchanged = false;
        std::vector<Words> local;   // whole trees: the nodes, read locally
    };
    AuthJob add_auth(Exchange& ex, const std::vector<const Tree*>& trees, const std::vector<u64>& indices) {
        AuthJob a;
        a.trees = trees;
        a.need = auth_node_indices(trees[0]->n_leaves, indices);
        if (a.need.empty()) return a;
        if (!trees[0]->split) {
            GatherBatch batch;
            for (const Tree* t : trees) batch.add(t->nodes.ptr(), 5, a.need);
            batch.run(c);
            for (size_t t = 0; t < trees.size(); t++) a.local.push_back(batch.jobs[t].out);
            return a;
        }
        // node k of the whole tree sits at depth d = floor(log2 k); below the subtree roots (depth >= log2 R) its position
        // pos = k - 2^d within the level selects subtree pos >> (d - log2 R), where it is node 2^(d - log2 R) + the low bits
        unsigned log_r = 0;
        while ((1ull << log_r) < R) log_r++;
        std::vector<u64> owner(a.need.size()), my_local;
        for (size_t k = 0; k < a.need.size(); k++) {
            const u64 node = a.need[k];
            if (node < 2 * R) {
                owner[k] = R;  // the top tree, on every rank
                continue;
            }
            const unsigned d = bit_length(node) - 1, dd = d - log_r;
            const u64 pos = node - (1ull << d);
            owner[k] = pos >> dd;
            if (owner[k] == me) my_local.push_back((1ull << dd) + (pos & ((1ull << dd) - 1)));
        }
        const size_t T = trees.size();
        Words mine(my_local.size() * T * 5);
        if (!my_local.empty()) {
            GatherBatch batch;
            for (const Tree* t : trees) batch.add(t->nodes.ptr(), 5, my_local);
            batch.run(c);
            for (size_t k = 0; k < my_local.size(); k++)
                for (size_t t = 0; t < T; t++) std::copy_n(batch.jobs[t].out.begin() + 5 * k, 5, mine.begin() + (k * T + t) * 5);
        }
        a.job = ex.add(std::move(owner), (uint32_t)(5 * T), std::move(mine));
        a.exchanged = true;
        return a;
    }
    std::vector<Words> take_auth(const Exchange& ex, const AuthJob& a) {   // after run(ex): per tree, the nodes in the order of `need`
        const size_t T = a.trees.size();
        std::vector<Words> out(T, Words(a.need.size() * 5));
        if (a.need.empty()) return out;
        if (!a.exchanged) return a.local;
        const Exchange::Job& job = ex.jobs[a.job];
        for (size_t k = 0; k < a.need.size(); k++)
            for (size_t t = 0; t < T; t++) {
                const u64* src = job.owner[k] == R ? &a.trees[t]->top[5 * a.need[k]] : &job.out[(k * T + t) * 5];
                std::copy_n(src, 5, out[t].begin() + 5 * k);
            }
        return out;
    }

    // ---------------------------------------------------------------------------------------------- master tables
    ArithmeticDomain group(const ArithmeticDomain& d, u64 pass) const { return coset_group(d, me + R * pass, R * P); }

    // hash_all_ldt_domain_rows + merkle_tree (master_table.rs:443-503) -> the tree over all L rows
    Tree commit_master_table(MasterTable& mt, const char* what) {
        const u64 L = p.ldt.length, local_rows = L / R, pass_rows = local_rows / P;
        DeviceBuffer digests(c, 5 * local_rows);
        if (P == 1) {
            c.check(tvm_hash_rows(c.raw(), mt.table(), local_rows, digests.ptr()), "tvm_hash_rows");
        } else {
            DeviceBuffer part(c, 5 * pass_rows);
            for (u64 s = 0; s < P; s++) {  // nothing is cached: extend, hash, drop (master_table.rs:470-503)
                const ArithmeticDomain g = group(p.ldt, s);
                mt.set_domains(g, g);
                mt.maybe_low_degree_extend_all_columns();
                c.check(tvm_hash_rows(c.raw(), mt.table(), pass_rows, part.ptr()), "tvm_hash_rows");
                scatter(part.ptr(), 5, pass_rows, P, s, digests.ptr());
            }
            mt.clear_cache();
        }
        return tree_from_local(std::move(digests), L, false, what);
    }

    // The valid-trace AIR over the ranks.  On a valid trace the constraint quotients are polynomials of known
    // length: class H ("half") has fewer than 4N coefficients, class QT (the "quarter" and "three cosets" classes together) fewer
    // than 3N, so the values of H on ANY four cosets of the trace domain and of QT on any three determine them -- the single-GPU
    // path takes the even cosets for both, which in a coset sharding all sit on the even ranks.  Here the seven (class, coset)
    // evaluations are dealt to the ranks by load (a rank can only evaluate on cosets it holds), every rank also evaluates the
    // four full-domain constraints on its own cosets, ONE all-gather moves all values, and every rank rebuilds the quotient
    // codeword: interpolate per coset, invert the Vandermonde matrix of the cosets' X^N per coefficient, evaluate the sum on the
    // whole domain, add the full-domain class.  The same field elements as tvm_all_quotients_combined yields (valid trace), at
    // about 1/R of its work per rank instead of 1/R of the EXACT evaluation.  Returns an empty buffer when it does not apply.
    DeviceBuffer quotient_codeword_by_classes(const std::vector<Xfe>& challenges, const std::vector<Xfe>& weights) {
        const u64 Q = p.quotient.length, N = p.trace.length, X = Q / N;
        if (!comm || R < 2 || P != 1 || Q != p.ldt.length || X < 4 || X % R) return DeviceBuffer();
        uint32_t need[4] = {0, 0, 0, 0};
        c.check(tvm_air_class_cosets(sp.main_.table(), sp.aux_.table(), p.trace.c(), need), "tvm_air_class_cosets");
        if (need[1] != 4 || need[2] != 2 || need[3] != 3) return DeviceBuffer();   // the degree bounds do not hold (h large against N)
        struct Task {
            uint32_t mask;
            u64 coset;   // of the quotient domain: the points coset + X j
        };
        const struct { uint32_t mask; unsigned cosets, weight; } classes[2] = {{TVM_AIR_CLASS_HALF, 4, 5}, {TVM_AIR_CLASS_QUARTER | TVM_AIR_CLASS_THREE, 3, 2}};
        std::vector<Task> tasks;
        std::vector<u64> load(R, 0);
        for (const auto& cl : classes) {
            std::vector<bool> used(X, false);
            for (unsigned i = 0; i < cl.cosets; i++) {   // the unused coset whose owner has the least work so far
                u64 best = X;
                for (u64 k = 0; k < X; k++)
                    if (!used[k] && (best == X || load[k % R] < load[best % R])) best = k;
                used[best] = true;
                load[best % R] += cl.weight;
                tasks.push_back(Task{cl.mask, best});
            }
        }
        std::vector<u64> n_tasks(R, 0), slot_of(tasks.size());
        for (size_t t = 0; t < tasks.size(); t++) slot_of[t] = n_tasks[tasks[t].coset % R]++;
        const u64 task_slots = *std::max_element(n_tasks.begin(), n_tasks.end()), own = X / R, slots = task_slots + own, slot_words = 3 * N;
        const ArithmeticDomain ldt_rank = coset_group(p.ldt, me, R);
        DeviceBuffer send(c, slots * slot_words), recv(c, R * slots * slot_words);
        for (u64 sl = n_tasks[me]; sl < task_slots; sl++)   // (unused slots travel as zeros: an evaluation of the zero polynomial)
            c.check(tvm_evaluate(c.raw(), 3, nullptr, 0, tvm_domain{to_mont(1), to_mont(1), N}, send.ptr() + sl * slot_words), "zero fill");
        for (size_t t = 0; t < tasks.size(); t++)
            if (tasks[t].coset % R == me)
                c.check(tvm_air_class_values(c.raw(), sp.main_.table(), sp.aux_.table(), p.trace.c(), ldt_rank.c(), (uint32_t)(tasks[t].coset / R),
                                             tasks[t].mask, challenges[0].c, weights[0].c, send.ptr() + slot_of[t] * slot_words), "tvm_air_class_values");
        for (u64 j = 0; j < own; j++)
            c.check(tvm_air_class_values(c.raw(), sp.main_.table(), sp.aux_.table(), p.trace.c(), ldt_rank.c(), (uint32_t)j, TVM_AIR_CLASS_FULL,
                                         challenges[0].c, weights[0].c, send.ptr() + (task_slots + j) * slot_words), "tvm_air_class_values");
        comm_check(comm->all_gather(comm->self, c.raw(), send.ptr(), recv.ptr(), slots * slot_words), "quotient class values");
        count("quotient class values", slots * slot_words * 8 * (R - 1));
        auto slot = [&](u64 rank, u64 s) { return recv.ptr() + (rank * slots + s) * slot_words; };
        DeviceBuffer coeffs(c, 4 * N * 3);
        c.check(tvm_evaluate(c.raw(), 3, nullptr, 0, tvm_domain{to_mont(1), to_mont(1), 4 * N}, coeffs.ptr()), "zero fill");
        size_t at = 0;
        for (const auto& cl : classes) {
            u64 offsets[4];
            const u64* values[4];
            for (unsigned i = 0; i < cl.cosets; i++, at++) {
                offsets[i] = p.quotient.value(tasks[at].coset);
                values[i] = slot(tasks[at].coset % R, slot_of[at]);
            }
            c.check(tvm_coset_values_to_coefficients(c.raw(), p.trace.c(), cl.cosets, offsets, values, coeffs.ptr()), "tvm_coset_values_to_coefficients");
        }
        DeviceBuffer out = p.quotient.evaluate(c, coeffs.ptr(), 4 * N, 3);
        DeviceBuffer full(c, 3 * Q);
        for (u64 k = 0; k < X; k++) scatter(slot(k % R, task_slots + k / R), 3, N, X, k, full.ptr());
        c.check(tvm_xfe_add_assign(c.raw(), out.ptr(), full.ptr(), Q), "tvm_xfe_add_assign");
        return out;
    }

    // all_quotients_combined over the quotient domain (master_table.rs:1264-1363): every group on its own rows -> all rows
    DeviceBuffer quotient_codeword(const std::vector<Xfe>& challenges, const std::vector<Xfe>& weights) {
        if (sp.assume_valid_trace) {
            DeviceBuffer by_classes = quotient_codeword_by_classes(challenges, weights);
            if (by_classes.ptr()) return by_classes;
        }
        const u64 Q = p.quotient.length, local_rows = Q / R, pass_rows = local_rows / P;
        const bool cached = P == 1 && Q == p.ldt.length;  // the quotient rows are the rows of the cached tables
        DeviceBuffer local(c, 3 * local_rows), part;
        if (P > 1) part = DeviceBuffer(c, 3 * pass_rows);
        // With a quotient domain shorter than the LDT domain (LDT expansion 16: the quotient rows are the LDT cosets
        // k = 0 (mod 4), which sit on a quarter of the ranks) every rank extends the traces once more onto ITS share of the
        // quotient domain, so that the AIR stays spread evenly over the ranks; the cached LDT-domain tables stay.
        std::unique_ptr<MasterTable> tmp_main, tmp_aux;
        MasterTable *m = &sp.main_, *a = &sp.aux_;
        if (!cached && P == 1) {
            tmp_main.reset(new MasterTable(sp.main_.sibling()));
            tmp_aux.reset(new MasterTable(sp.aux_.sibling()));
            m = tmp_main.get(), a = tmp_aux.get();
        }
        if (sp.assume_valid_trace) c.check(tvm_ctx_set_option(c.raw(), TVM_OPTION_AIR_VALID_TRACE, 1), "tvm_ctx_set_option");
        int32_t status = TVM_OK;
        try {
            for (u64 s = 0; s < P && status == TVM_OK; s++) {
                const ArithmeticDomain g = group(p.quotient, s);
                if (!cached) {
                    m->set_domains(g, g);
                    m->maybe_low_degree_extend_all_columns();
                    a->set_domains(g, g);
                    a->maybe_low_degree_extend_all_columns();
                }
                status = tvm_all_quotients_combined(c.raw(), m->table(), a->table(), p.trace.c(), g.c(), challenges[0].c, weights[0].c,
                                                    P == 1 ? local.ptr() : part.ptr());
                if (status == TVM_OK && P > 1) scatter(part.ptr(), 3, pass_rows, P, s, local.ptr());
            }
        } catch (...) {
            if (sp.assume_valid_trace) (void)tvm_ctx_set_option(c.raw(), TVM_OPTION_AIR_VALID_TRACE, 0);
            throw;
        }
        if (sp.assume_valid_trace) (void)tvm_ctx_set_option(c.raw(), TVM_OPTION_AIR_VALID_TRACE, 0);
        c.check(status, "tvm_all_quotients_combined");
        if (!cached) m->clear_cache(), a->clear_cache();
        return gather_rows(std::move(local), local_rows, 3, "quotient codeword");
    }

    // out_of_domain_row at several indeterminates (master_table.rs:348-390), the columns split evenly over the ranks (the
    // traces are replicated) -> [n_points][n_cols][3]
    Words out_of_domain_rows(const MasterTable& mt, const std::vector<Xfe>& points) {
        if (!comm) return mt.out_of_domain_rows(points);
        const u64 n_cols = mt.n_cols(), per = (n_cols + R - 1) / R, n_pts = points.size();
        const u64 c0 = std::min(me * per, n_cols), c1 = std::min(c0 + per, n_cols);
        const Words mine = mt.out_of_domain_rows(points, c0, c1 - c0);  // [n_points][c1 - c0][3]
        Words by_column((c1 - c0) * n_pts * 3);
        for (u64 pt = 0; pt < n_pts; pt++)
            for (u64 col = 0; col < c1 - c0; col++) std::copy_n(&mine[(pt * (c1 - c0) + col) * 3], 3, &by_column[(col * n_pts + pt) * 3]);
        const std::vector<Words> all = all_gather_host(by_column, per * n_pts * 3, "out-of-domain rows");
        Words out(n_pts * n_cols * 3);
        for (u64 col = 0; col < n_cols; col++)
            for (u64 pt = 0; pt < n_pts; pt++) std::copy_n(&all[col / per][((col % per) * n_pts + pt) * 3], 3, &out[(pt * n_cols + col) * 3]);
        return out;
    }

    // reveal_rows (master_table.rs:548-609): row i of the extended table lives on rank i % R as local row a = i / R, and
    // in pass a % P of that rank as row a / P
    size_t add_master_rows(Exchange& ex, MasterTable& mt, const std::vector<u64>& indices) {
        const u64 width = mt.n_cols() * mt.field_kind(), local_rows = p.ldt.length / R;
        auto fetch = [&](const std::vector<u64>& local) {
            if (P == 1) return mt.reveal_rows_of(local, local_rows);
            Words rows(local.size() * width);
            for (u64 s = 0; s < P; s++) {
                std::vector<u64> in_pass, at;
                for (size_t k = 0; k < local.size(); k++)
                    if (local[k] % P == s) in_pass.push_back(local[k] / P), at.push_back(k);
                if (in_pass.empty()) continue;
                const ArithmeticDomain g = group(p.ldt, s);
                mt.set_domains(g, g);
                mt.maybe_low_degree_extend_all_columns();
                const Words got = mt.reveal_rows_of(in_pass, local_rows / P);
                for (size_t k = 0; k < at.size(); k++) std::copy_n(&got[k * width], width, &rows[at[k] * width]);
            }
            mt.clear_cache();
            return rows;
        };
        return add_distributed(ex, indices, (uint32_t)width, fetch);
    }

    // ---------------------------------------------------------------------------------------------- FRI
    // Fri::prove (fri.rs:212-319) on a codeword distributed by residue class.  With x_i = o w^i the fold (fri.rs:349-366)
    // pairs i with i + n/2, which share a residue mod R as long as R divides n/2: rank r folds ITS elements, and they are
    // the residue-r elements of the next codeword -- split_and_fold on the rank's domain (o w^r, w^R, n/R), whose square
    // is the rank's domain of the next round.  The tree of a round needs contiguous leaf ranges: one all-to-all per round.
    // From the first round whose tree is built whole, the codeword is gathered once and the commit phase continues with
    // the sponge on the device (tvm_fri_commit_phase), replicated.
    std::vector<u64> fri(DeviceBuffer&& combination_local) {
        struct Round {
            ArithmeticDomain dom;
            const u64* cw = nullptr;  // distributed: this rank's elements; else the whole codeword
            bool distributed = false;
            Tree tree;
        };
        std::vector<Round> rounds;
        std::vector<DeviceBuffer> owned;
        owned.push_back(std::move(combination_local));
        ArithmeticDomain dom = p.ldt, local_dom = coset_group(p.ldt, me, R);
        unsigned r = 0;
        bool folded_past_last = false;
        for (; r <= p.fri_rounds && splits(dom.length); r++) {
            Round round;
            round.dom = dom;
            round.cw = owned.back().ptr();
            round.distributed = true;
            {
                DeviceBuffer copy(c, 3 * local_dom.length);  // (tree_from_local consumes its argument)
                c.check(tvm_memcpy_d2d(c.raw(), copy.ptr(), round.cw, 3 * local_dom.length * 8), "tvm_memcpy_d2d");
                round.tree = tree_from_local(std::move(copy), dom.length, true, "FRI codeword");
            }
            ps.enqueue("fri root " + std::to_string(r), round.tree.root().data(), 5);
            rounds.push_back(std::move(round));
            if (r == p.fri_rounds) {
                folded_past_last = true;
                r++;
                break;
            }
            const Xfe challenge = ps.sample_scalars(1)[0];
            DeviceBuffer next(c, 3 * (local_dom.length / 2));
            c.check(tvm_fri_split_and_fold(c.raw(), owned.back().ptr(), local_dom.c(), challenge.c, next.ptr()), "tvm_fri_split_and_fold");
            owned.push_back(std::move(next));
            dom = dom.pow(2);
            local_dom = local_dom.pow(2);
        }
        // the codeword of round r (or, when every round's tree was split, the last codeword) in row order on every rank
        const u64* cw;
        {
            DeviceBuffer local = std::move(owned.back());
            owned.pop_back();
            const bool keep = !rounds.empty() && rounds.back().distributed && rounds.back().cw == local.ptr();
            if (comm && keep) {  // the last split round still answers queries from the distributed codeword
                DeviceBuffer copy(c, 3 * local_dom.length);
                c.check(tvm_memcpy_d2d(c.raw(), copy.ptr(), local.ptr(), 3 * local_dom.length * 8), "tvm_memcpy_d2d");
                owned.push_back(std::move(local));
                local = std::move(copy);
            }
            owned.push_back(gather_rows(std::move(local), local_dom.length, 3, "FRI codeword (gathered)"));
            cw = owned.back().ptr();
        }
        if (!folded_past_last) {
            const unsigned left = p.fri_rounds - r;  // folds still to do; trees for rounds r .. fri_rounds
            std::vector<u64*> d_cw, d_nodes;
            ArithmeticDomain d = dom;
            const size_t first = rounds.size();
            for (unsigned k = 0; k <= left; k++) {
                Round round;
                round.dom = d;
                round.tree.n_leaves = d.length;
                round.tree.nodes = DeviceBuffer(c, 10 * d.length);
                d_nodes.push_back(round.tree.nodes.ptr());
                rounds.push_back(std::move(round));
                if (k == left) break;
                owned.emplace_back(c, d.length / 2 * 3);
                d_cw.push_back(owned.back().ptr());
                d = d.pow(2);
            }
            Words roots(5 * (left + 1)), challenges(3 * (size_t)left + 1);
            c.check(tvm_fri_commit_phase(c.raw(), cw, dom.c(), left, ps.sponge_state(), d_cw.data(), d_nodes.data(), roots.data(),
                                         challenges.data()), "tvm_fri_commit_phase");
            for (unsigned k = 0; k <= left; k++) {
                Round& round = rounds[first + k];
                round.cw = k == 0 ? cw : d_cw[k - 1];
                round.tree.top.assign(10, 0);
                std::copy_n(&roots[5 * k], 5, round.tree.top.begin() + 5);
                ps.enqueue("fri root " + std::to_string(r + k), &roots[5 * k], 5);
                if (k == left) break;
                const Xfe challenge = ps.sample_scalars(1)[0];
                if (std::memcmp(challenge.c, &challenges[3 * k], 3 * sizeof(u64)) != 0)
                    throw Error(TVM_ERR_DEVICE, "the device's Fiat-Shamir sponge and the host's disagree on a FRI folding challenge");
            }
            cw = rounds.back().cw;
            dom = rounds.back().dom;
        }
        Words last(dom.length * 3);
        c.check(tvm_memcpy_d2h(c.raw(), last.data(), cw, last.size() * sizeof(u64)), "last codeword");
        ps.enqueue("fri last codeword", last.data(), last.size());
        const DeviceBuffer last_poly_d = ArithmeticDomain::of_length(dom.length).interpolate(c, cw, 3);
        const Words last_poly = last_poly_d.download(0, dom.length * 3);
        ps.enqueue("fri last polynomial", last_poly.data(), last_poly.size());
        const std::vector<u64> a_indices = ps.sample_indices(p.ldt.length, p.num_collinearity_checks);
        // the responses of all rounds in ONE exchange (their order in the proof stream is fixed below)
        Exchange ex;
        struct Response {
            size_t round;
            bool distributed;
            size_t leaves_job;
            Words leaves;
            AuthJob auth;
        };
        std::vector<Response> responses;
        for (size_t k = 0; k < rounds.size(); k++) {
            const Round& round = rounds[k];
            std::vector<u64> b_idx;
            for (u64 i : a_indices) b_idx.push_back((i % round.dom.length + round.dom.length / 2) % round.dom.length);
            for (int which = (k == 0 ? 0 : 1); which < 2; which++) {
                if (which == 1 && k == rounds.size() - 1) continue;
                const std::vector<u64>& ix = which == 0 ? a_indices : b_idx;
                auto fetch = [&](const std::vector<u64>& at) {
                    Words out(at.size() * 3);
                    c.check(tvm_gather_elements(c.raw(), round.cw, 3, at.data(), at.size(), out.data()), "tvm_gather_elements");
                    return out;
                };
                Response q{k, round.distributed, 0, {}, {}};
                if (round.distributed) q.leaves_job = add_distributed(ex, ix, 3, fetch);
                else q.leaves = fetch(ix);
                q.auth = add_auth(ex, {&round.tree}, ix);
                responses.push_back(std::move(q));
            }
        }
        run(ex, "FRI responses and authentication nodes");
        for (const Response& q : responses) {
            const Words& leaves = q.distributed ? ex.jobs[q.leaves_job].out : q.leaves;
            const Words auth = take_auth(ex, q.auth)[0];
            ps.enqueue("fri response " + std::to_string(q.round), leaves.data(), leaves.size());
            ps.enqueue("fri auth " + std::to_string(q.round), auth.data(), auth.size());
        }
        (void)ps.sample_scalars(1);
        return a_indices;
    }

    // ---------------------------------------------------------------------------------------------- the proof
    ProofStream prove() {
        sp.stage_ms.clear();
        sp.exchanges.clear();
        sp.split_trees_built = 0;
        ps.alter_fiat_shamir_state_with(sp.claim_.encode());  // stark.rs:336-339
        {
            const u64 log2_padded_height = to_mont(bit_length(p.padded_height) - 1);  // stark.rs:354
            ps.enqueue("log2 padded height", &log2_padded_height, 1);
        }
        const u64 L = p.ldt.length;
        const ArithmeticDomain short_dom = p.ldt.length <= p.quotient.length ? p.ldt : p.quotient;
        const ArithmeticDomain ldt_rank = coset_group(p.ldt, me, R), short_rank = coset_group(short_dom, me, R);
        const u64 zeta = to_mont(3);  // Stark::ZETA, stark.rs:1801
        auto enqueue_xfes = [&](const char* name, const std::vector<Xfe>& v) { ps.enqueue(name, v[0].c, 3 * v.size()); };
        MasterTable &main = sp.main_, &aux = sp.aux_;

        // 4-6: main table LDE, Merkle tree, challenges  (stark.rs:367-377)
        // (TVMH_OPTION_COLUMN_SPLIT: the inverse transforms split by columns, the coefficients exchanged -- see extend_table)
        const unsigned column_chunks = comm && R > 1 && P == 1 ? (unsigned)std::min<u64>(tvmh_get_option(TVMH_OPTION_COLUMN_SPLIT), 16) : 0;
        auto extend_table = [&](MasterTable& mt, const char* what) {
            mt.set_domains(ldt_rank, ldt_rank);
            if (column_chunks) mt.low_degree_extend_over(comm, column_chunks, [&](u64 bytes) { count(what, bytes); });
            else mt.maybe_low_degree_extend_all_columns();
        };
        mark("main LDE");
        if (P == 1) extend_table(main, "main coefficients");
        mark("main Merkle");
        const Tree main_tree = commit_master_table(main, "main leaf digests");
        ps.enqueue("main root", main_tree.root().data(), 5);
        const std::vector<Xfe> challenges = derive_challenges(ps.sample_scalars(NUM_SAMPLED_CHALLENGES), sp.claim_);
        mark("extend");
        if (sp.extend) sp.extend(challenges);  // MasterMainTable::extend (stark.rs:379-381), replicated

        // 8-9: aux table
        mark("aux LDE");
        if (P == 1) extend_table(aux, "aux coefficients");
        mark("aux Merkle");
        const Tree aux_tree = commit_master_table(aux, "aux leaf digests");
        ps.enqueue("aux root", aux_tree.root().data(), 5);
        const std::vector<Xfe> quotient_weights = xfe_powers(ps.sample_scalars(1)[0], 0, TVM_NUM_QUOTIENT_WEIGHTS);

        // 10: quotient codeword, segments, randomization  (stark.rs:405-423).  The segment polynomials need the whole
        // codeword (one interpolation, replicated); the segment TABLE is evaluated on this rank's rows only.
        mark("AIR quotients");
        DeviceBuffer quot = quotient_codeword(challenges, quotient_weights);
        mark("quotient segments LDE");
        const u64 poly_len = std::max<u64>(p.quotient.length / 4, sp.quotient_randomizer_.size());
        DeviceBuffer polys(c, 5 * poly_len * 3);
        TableGuard seg{c};
        c.check(tvm_quotient_segments(c.raw(), quot.ptr(), p.quotient.c(), ldt_rank.c(), sp.quotient_randomizer_.data()->c,
                                      sp.quotient_randomizer_.size(), zeta, &seg.t, polys.ptr(), poly_len), "tvm_quotient_segments");
        quot.reset();
        // 12: quotient Merkle tree  (stark.rs:425-446)
        mark("quotient Merkle");
        Tree quot_tree;
        {
            DeviceBuffer digests(c, 5 * ldt_rank.length);
            c.check(tvm_hash_rows(c.raw(), seg.t, ldt_rank.length, digests.ptr()), "tvm_hash_rows");
            quot_tree = tree_from_local(std::move(digests), L, false, "quotient leaf digests");
        }
        ps.enqueue("quot root", quot_tree.root().data(), 5);

        // 13: out-of-domain rows  (stark.rs:450-495)
        mark("out-of-domain rows");
        const Xfe alpha = ps.sample_scalars(1)[0];
        const Xfe alpha_next = xfe_scale(alpha, p.trace.generator);
        const Words ood_main = out_of_domain_rows(main, {alpha, alpha_next});
        const Words ood_aux = out_of_domain_rows(aux, {alpha, alpha_next});
        const Xfe a4 = xfe_powers(alpha, 4, 1)[0];
        const Xfe za4 = xfe_powers(xfe_scale(alpha, zeta), 4, 1)[0];
        Xfe seg_ood[5][2];
        {
            const Xfe pts[2] = {a4, za4};   // the five segment polynomials at both points: one round trip
            c.check(tvm_evaluate_polys_at_points(c.raw(), polys.ptr(), poly_len, poly_len, 5, pts[0].c, 2, seg_ood[0][0].c),
                    "tvm_evaluate_polys_at_points");
        }
        ps.enqueue("ood main", ood_main.data(), NUM_MAIN * 3);
        ps.enqueue("ood aux", ood_aux.data(), NUM_AUX * 3);
        ps.enqueue("ood main next", ood_main.data() + NUM_MAIN * 3, NUM_MAIN * 3);
        ps.enqueue("ood aux next", ood_aux.data() + NUM_AUX * 3, NUM_AUX * 3);
        enqueue_xfes("ood quot p", {seg_ood[0][0], seg_ood[1][0], seg_ood[2][0], seg_ood[3][0]});
        enqueue_xfes("ood quot r", {seg_ood[1][1], seg_ood[2][1], seg_ood[3][1], seg_ood[4][1]});

        // 14-15: combination weights, linear combinations  (stark.rs:497-543), on this rank's rows of the short domain
        mark("linear combination");
        const std::vector<Xfe> w3 = ps.sample_scalars(3);
        const std::vector<Xfe> weights_ma = xfe_powers(w3[0], 0, NUM_MAIN + NUM_AUX);
        const std::vector<Xfe> weights_q = xfe_powers(w3[1], 0, 5);
        const std::vector<Xfe> weights_d = xfe_powers(w3[2], 0, 4);
        DeviceBuffer comb = main.weighted_sum_of_columns(&weights_ma[0]);
        {
            const DeviceBuffer comb_aux = aux.weighted_sum_of_columns(&weights_ma[NUM_MAIN]);
            c.check(tvm_xfe_add_assign(c.raw(), comb.ptr(), comb_aux.ptr(), 2 * p.trace.length), "tvm_xfe_add_assign");
        }
        const u64 n_comb = p.trace.length + p.h;
        const DeviceBuffer main_aux_codeword = short_rank.evaluate(c, comb.ptr(), n_comb, 3);
        std::vector<Xfe> wp = weights_q, wr = weights_q;
        wp[4] = Xfe{{0, 0, 0}};
        wr[0] = Xfe{{0, 0, 0}};
        // values of the P and R polynomials (stark.rs:520-540) on this rank's rows of the short domain.  When that is the LDT
        // domain they are row-wise combinations of the rank's segment table.  When the quotient domain is the short one, its
        // row s is LDT row s |LDT| / |quotient|, which lives on rank (s |LDT| / |quotient|) mod R -- not on the rank that owns
        // short row s -- so the two polynomials are formed from the segment polynomials and evaluated on the rank's rows.
        DeviceBuffer cw_p, cw_r;
        if (short_dom.length == L) {
            cw_p = DeviceBuffer(c, short_rank.length * 3);
            cw_r = DeviceBuffer(c, short_rank.length * 3);
            c.check(tvm_table_linear_combination(c.raw(), seg.t, short_rank.length, wp[0].c, cw_p.ptr()), "tvm_table_linear_combination");
            c.check(tvm_table_linear_combination(c.raw(), seg.t, short_rank.length, wr[0].c, cw_r.ptr()), "tvm_table_linear_combination");
        } else {
            DeviceBuffer poly(c, poly_len * 3);
            c.check(tvm_xfe_linear_combination(c.raw(), polys.ptr(), 5, poly_len, poly_len, wp[0].c, poly.ptr()), "tvm_xfe_linear_combination");
            cw_p = short_rank.evaluate(c, poly.ptr(), poly_len, 3);
            c.check(tvm_xfe_linear_combination(c.raw(), polys.ptr(), 5, poly_len, poly_len, wr[0].c, poly.ptr()), "tvm_xfe_linear_combination");
            cw_r = short_rank.evaluate(c, poly.ptr(), poly_len, 3);
        }
        Xfe ma_values[2];
        {
            const Xfe pts[2] = {alpha, alpha_next};
            c.check(tvm_evaluate_at_points(c.raw(), comb.ptr(), n_comb, pts[0].c, 2, ma_values[0].c), "tvm_evaluate_at_points");
        }
        Xfe p_value{{0, 0, 0}}, r_value{{0, 0, 0}};
        for (int k = 0; k < 4; k++) p_value = xfe_add(p_value, xfe_mul(weights_q[k], seg_ood[k][0]));
        for (int k = 1; k < 5; k++) r_value = xfe_add(r_value, xfe_mul(weights_q[k], seg_ood[k][1]));

        // 16: DEEP  (stark.rs:545-639), row-local
        mark("DEEP");
        DeviceBuffer combination(c, short_rank.length * 3);
        {
            const u64* cws[4] = {main_aux_codeword.ptr(), main_aux_codeword.ptr(), cw_p.ptr(), cw_r.ptr()};
            const Xfe points[4] = {alpha, alpha_next, a4, za4}, values[4] = {ma_values[0], ma_values[1], p_value, r_value};
            c.check(tvm_deep_codeword(c.raw(), 4, cws, short_rank.c(), points[0].c, values[0].c, weights_d[0].c, combination.ptr()),
                    "tvm_deep_codeword");
        }
        cw_p.reset();
        cw_r.reset();
        comb.reset();
        if (short_dom.length != L) {  // stark.rs:629-639: the quotient domain was the short one -- extend to the LDT domain
            const DeviceBuffer whole = gather_rows(std::move(combination), short_rank.length, 3, "combination codeword (short domain)");
            const DeviceBuffer coeffs = p.quotient.interpolate(c, whole.ptr(), 3);
            combination = ldt_rank.evaluate(c, coeffs.ptr(), p.quotient.length, 3);
        }

        // 17: the low-degree test  (stark.rs:641-663)
        mark(p.use_stir ? "STIR" : "FRI");
        std::vector<u64> a_indices;
        if (p.use_stir) {  // Stir::prove on the whole codeword, replicated
            const DeviceBuffer whole = gather_rows(std::move(combination), ldt_rank.length, 3, "combination codeword");
            a_indices = p.stir.prove(c, whole.ptr(), ps);
        } else {
            a_indices = fri(std::move(combination));
        }

        // 18: the out-of-domain point must not collide with a revealed in-domain point  (stark.rs:645-663)
        if (a4.c[1] == 0 && a4.c[2] == 0) {
            const u64 other = mont_mul(a4.c[0], mont_pow(zeta, 4));
            for (u64 i : a_indices) {
                const u64 x = p.ldt.value(i);
                if (x == a4.c[0] || x == other) throw Error(TVM_ERR_INVALID_ARGUMENT, "ZeroKnowledgeViolation (stark.rs:645-663)");
            }
        }

        // 19: open the trace leafs  (stark.rs:665-716)
        mark("open trace leafs");
        {
            Exchange ex;   // the rows of the three tables and the authentication nodes of the three trees: one exchange
            const AuthJob auth_job = add_auth(ex, {&main_tree, &aux_tree, &quot_tree}, a_indices);
            const size_t main_job = add_master_rows(ex, main, a_indices), aux_job = add_master_rows(ex, aux, a_indices);
            auto fetch = [&](const std::vector<u64>& local) {
                Words rows(local.size() * 15);
                c.check(tvm_table_reveal_rows(c.raw(), seg.t, ldt_rank.length, local.data(), local.size(), rows.data()), "quotient rows");
                return rows;
            };
            const size_t quot_job = add_distributed(ex, a_indices, 15, fetch);
            run(ex, "opened rows and authentication nodes");
            const std::vector<Words> auth = take_auth(ex, auth_job);
            const Words &main_rows = ex.jobs[main_job].out, &aux_rows = ex.jobs[aux_job].out, &qrows = ex.jobs[quot_job].out;
            ps.enqueue("main rows", main_rows.data(), main_rows.size());
            ps.enqueue("main auth", auth[0].data(), auth[0].size());
            ps.enqueue("aux rows", aux_rows.data(), aux_rows.size());
            ps.enqueue("aux auth", auth[1].data(), auth[1].size());
            ps.enqueue("quot rows", qrows.data(), qrows.size());
            ps.enqueue("quot auth", auth[2].data(), auth[2].size());
        }
        main.clear_cache();
        aux.clear_cache();
        c.check(tvm_sync(c.raw()), "tvm_sync");
        close_stage();
        stage.clear();
        return std::move(ps);
    }
};

ProofStream ShardedProver::prove() {
    ShardedRun run(*this);
    return run.prove();
}

std::string ShardedProver::stats_json() const {
    std::string s = "{\"rank\": " + std::to_string(comm_ ? comm_->rank : 0) + ", \"world\": " + std::to_string(comm_ ? comm_->world : 1) +
                    ", \"passes\": " + std::to_string(passes_) + ", \"split_trees_built\": " + std::to_string(split_trees_built) + ", \"stage_ms\": {";
    char buf[96];
    for (size_t k = 0; k < stage_ms.size(); k++) {
        std::snprintf(buf, sizeof buf, "%s\"%s\": %.3f", k ? ", " : "", stage_ms[k].first.c_str(), stage_ms[k].second);
        s += buf;
    }
    s += "}, \"exchanges\": {";
    for (size_t k = 0; k < exchanges.size(); k++)
        s += std::string(k ? ", " : "") + "\"" + exchanges[k].first + "\": {\"calls\": " + std::to_string(exchanges[k].second.first) +
             ", \"bytes_sent\": " + std::to_string(exchanges[k].second.second) + "}";
    return s + "}}";
}

namespace {
struct CommSession {  // the communicator's begin / end hooks around one proof of this rank (also when it fails)
    const tvmh_comm* comm;
    const Context& c;
    CommSession(const tvmh_comm* comm_, const Context& c_) : comm(comm_), c(c_) {
        if (comm && comm->begin) comm->begin(comm->self, c.raw());
    }
    ~CommSession() {
        if (comm && comm->end) comm->end(comm->self, c.raw());
    }
};
}  // namespace

namespace {
// The replicated trace-side tables of one proof, with a Context of their own: when the ranks of an in-process communicator share
// them (tvmh_comm::share) the group keeps the object beyond the call that built it, and the buffers must not point at a
// Context on that call's stack.
struct SharedTables {
    Context c;
    ExecutionTables t;
    SharedTables(tvm_ctx* raw, const StarkParameters& p, const tvm_aet& aet, const uint8_t seed[32], const std::function<void(const char*)>& lap)
        : c(raw), t(c, p, aet, seed, lap) {}
    static void drop(const void* self) { delete (const SharedTables*)self; }
};

// A conservative estimate of what one rank holds at the peak of a proof with `passes` passes (bytes): its traces, the cached
// extension of its share of the rows (main, aux, quotient segments), the intermediates of one 96-column chunk of the table
// extension, and the codewords / digests of the quotient and DEEP steps.
u64 estimated_bytes(const StarkParameters& p, u64 world, u64 passes) {
    const u64 n = p.trace.length, L = p.ldt.length, share = L / (world * passes), X = std::max<u64>(L / n / (world * passes), 1);
    const u64 row_words = NUM_MAIN + 3 * NUM_AUX;
    u64 words = row_words * n;                                   // the traces
    words += (passes == 1 ? row_words + 15 : row_words) * share; // the cached extension (coset-wise passes keep one group at a time)
    words += 96 * n * (1 + X);                                   // table extension: one chunk's intermediates
    words += 5 * (L / world) * 3 + 3 * p.quotient.length * 4;    // leaf digests and their tree; quotient codeword, segments, combination
    words += 10 * (L / world);                                   // the Merkle nodes over a rank's leaves (three trees, one at a time + tops)
    // + a quarter and 48 MB for the pool's size classes (2 MB granules), the successor blocks of the tables and the small buffers:
    // with a communicator an out-of-memory in the middle of a proof is fatal, one pass too many only costs time
    return words * 10 + ((u64)48 << 20);
}
}  // namespace

std::vector<u64> prove_execution_sharded(const Context& c, const StarkParameters& p, const tvmh_comm* comm, unsigned passes, const tvm_aet& aet,
                                         const Claim& claim, const uint8_t seed[32], bool profile, std::string* stats,
                                         u64 split_tree_min_leaves, unsigned policy_first_passes) {
    const u64 n = p.trace.length;
    const CommSession session(comm, c);
    // TVMH_OPTION_TRACE: host wall time of the phases of one proof on stderr (no stream synchronisation is added)
    const bool trace = tvmh_get_option(TVMH_OPTION_TRACE) != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[tvmh sharded] %-32s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const u64 world = comm ? comm->world : 1;
    // in-process ranks may use ONE copy of the replicated tables (triton_host.hpp: tvmh_comm::share)
    const bool shared = comm && comm->share && world > 1 && tvmh_get_option(TVMH_OPTION_SHARE_REPLICATED_TABLES);
    auto rendezvous = [&](uint32_t op, const void* mine, const void** out) {
        const int32_t status = comm->share(comm->self, c.raw(), op, mine, mine ? &SharedTables::drop : nullptr, out);
        if (status != TVM_OK) throw Error(status, "the communicator's share hook failed (a rank left the proof)");
    };
    auto attempt = [&](unsigned pass_count) {
        if (comm && comm->mark) comm->mark(comm->self, c.raw(), "trace tables (fill, pad, randomizers)");
        lap("entry");
        std::unique_ptr<SharedTables> own;
        const ExecutionTables* t = nullptr;
        if (shared) {
            // every rank has left the previous proof -> rank 0 drops that proof's tables, builds this one's and hands them over
            rendezvous(TVMH_SHARE_RELEASE, nullptr, nullptr);
            const void* obj = nullptr;
            if (comm->rank == 0) {
                own.reset(new SharedTables(c.raw(), p, aet, seed, [&](const char* what) { lap(what); }));
                c.check(tvm_sync(c.raw()), "tvm_sync");   // the other ranks read the tables on their own streams
                // Ownership passes to the group the moment it installs the pointer -- which the hook reports through *out even
                // when it then returns an error (a peer left between its two barriers): own must let go exactly then, or the group's
                // kept_drop and this unique_ptr would both free the tables.
                const int32_t handed = comm->share(comm->self, c.raw(), TVMH_SHARE_PUBLISH, own.get(), &SharedTables::drop, &obj);
                if (obj == own.get()) own.release();      // the group keeps it
                if (handed != TVM_OK) throw Error(handed, "the communicator's share hook failed (a rank left the proof)");
            } else {
                rendezvous(TVMH_SHARE_PUBLISH, nullptr, &obj);
            }
            t = &((const SharedTables*)obj)->t;
        } else {
            own.reset(new SharedTables(c.raw(), p, aet, seed, [&](const char* what) { lap(what); }));
            t = &own->t;
        }
        ShardedProver prover(c, p, comm, pass_count, t->main_trace.ptr(), t->main_rnd.ptr(), t->aux_trace.ptr(), t->aux_rnd.ptr(),
                             t->quotient_randomizer, claim);
        prover.assume_valid_trace = !tvmh_get_option(TVMH_OPTION_EXACT_AIR);
        prover.profile = profile;
        prover.split_tree_min_leaves = split_tree_min_leaves;
        prover.extend = [&](const std::vector<Xfe>& challenges) {
            if (!shared) return t->extend(c, n, challenges);
            if (comm->rank == 0) {   // (every rank derives the same challenges: the transcripts are identical)
                t->extend(c, n, challenges);
                c.check(tvm_sync(c.raw()), "tvm_sync");
            }
            rendezvous(TVMH_SHARE_BARRIER, nullptr, nullptr);
        };
        ProofStream stream = prover.prove();
        lap("prove (device drained)");
        std::vector<u64> proof = stream.proof();
        if (stats) *stats = prover.stats_json();
        lap("proof encoding");
        return proof;
    };
    struct Done {
        decltype(lap)& l;
        ~Done() { l("tables released, return"); }
    } done{lap};
    // A rank that fails in the middle of a proof will not join the collectives its peers are waiting in: say so to the communicator
    // (RCCL: ncclCommAbort; in-process: the waiting ranks leave with an error) before the error travels up.
    auto guarded_attempt = [&](unsigned pass_count) {
        try {
            return attempt(pass_count);
        } catch (...) {
            if (comm && world > 1 && comm->abort) comm->abort(comm->self);
            throw;
        }
    };
    if (passes) return guarded_attempt(passes);
    // The reference's memory policy (master_table.rs:268-271, stark.rs:730-768): try the cached extension; if the device (or
    // the context's memory limit) cannot hold it, start over coset by coset with as few passes as fit.  The transcript is
    // deterministic, so the restarted proof is the proof the cached path would have produced.
    const u64 expansion = p.ldt.length / p.trace.length;
    auto may_double = [&](unsigned pass_count) { return pass_count * 2 * world <= expansion && p.quotient.length == p.ldt.length; };
    if (world > 1) {
        // With a communicator the decision has to be COLLECTIVE and taken BEFORE the first collective of the proof: a rank that ran
        // out of memory on its own and restarted while its peers sat in an all-gather would issue mismatched collectives (the ranks'
        // free memory differs: other tenants, fragmentation, per-context limits).  So every rank takes the smallest pass count
        // whose estimated footprint fits what ITS context can still obtain (tvm_ctx_memory_info: device free + own cache, the
        // memory limit included), one all-gather makes the largest of them everybody's, and an out-of-memory during the proof
        // itself is an error (and an abort of the communicator), not a local retry.
        size_t available = 0;
        c.check(tvm_ctx_memory_info(c.raw(), &available, nullptr), "tvm_ctx_memory_info");
        unsigned mine = 1;
        while (estimated_bytes(p, world, mine) > available && may_double(mine)) mine *= 2;
        DeviceBuffer send(c, 1), recv(c, world);
        const u64 word = mine;
        c.check(tvm_memcpy_h2d(c.raw(), send.ptr(), &word, 8), "tvm_memcpy_h2d");
        const int32_t status = comm->all_gather(comm->self, c.raw(), send.ptr(), recv.ptr(), 1);
        if (status != TVM_OK) throw Error(status, "pass-count agreement: the communicator reported " + std::string(tvm_status_string(status)));
        const std::vector<u64> all = recv.download(0, world);
        return guarded_attempt((unsigned)*std::max_element(all.begin(), all.end()));
    }
    // One rank: the attempts themselves find out what fits (an out-of-memory unwinds, the pool goes back to the driver, the proof starts
    // over with more passes) -- but not blindly.  (a) A pass count whose footprint cannot fit even by the plain estimate (the
    // conservative one without its quarter of headroom) is not attempted: at 2^23 rows the cached path would allocate for seconds
    // before it fails.  (b) A coset-wise pass count has to fit by the CONSERVATIVE estimate: the passes allocate and release their
    // group's tables again and again, and on a device that the working set just about fills the pool cannot keep any of them (measured,
    // 2^23 rows on one MI355X: 2 passes "fit" and take 27 s per proof, hipMalloc of 100-GiB blocks in every pass; profiles/r05_t_*).
    size_t available = 0;
    c.check(tvm_ctx_memory_info(c.raw(), &available, nullptr), "tvm_ctx_memory_info");
    auto fits = [&](unsigned pass_count) {
        const u64 conservative = estimated_bytes(p, 1, pass_count);
        return (pass_count == 1 ? conservative / 10 * 8 : conservative) <= available;
    };
    unsigned pass_count = policy_first_passes ? policy_first_passes : 1;   // (tvmh_prove_execution arrives here after ITS cached path failed: 2)
    while (!fits(pass_count) && may_double(pass_count)) pass_count *= 2;
    for (;;) {
        try {
            return attempt(pass_count);
        } catch (const Error& e) {
            if (e.status != TVM_ERR_OUT_OF_MEMORY || !may_double(pass_count)) throw;
        }
        (void)tvm_ctx_trim(c.raw());  // the failed attempt's buffers went back to the pool while unwinding: give them to the driver
        do pass_count *= 2; while (!fits(pass_count) && may_double(pass_count));
    }
}

// ------------------------------------------------------------------------------------------------ local communicators
namespace {
struct LocalGroup {
    uint32_t world;
    bool lockstep;
    std::mutex m;
    std::condition_variable cv;
    uint64_t arrived = 0, generation = 0;
    std::vector<const uint64_t*> send;
    // lockstep: whose turn it is to compute (world = nobody, everybody is inside a collective), per-rank stage clocks
    uint32_t turn = 0;
    std::vector<std::string> stage;
    std::vector<std::chrono::steady_clock::time_point> since;
    std::vector<std::string> stage_order;
    std::map<std::string, std::vector<double>> ms;
    std::vector<tvmh_comm> comms;
    struct Member {
        LocalGroup* group;
        uint32_t rank;
    };
    std::vector<Member> members;

    // tvmh_comm::share: the object rank 0 published (the replicated tables of the current proof), dropped at the next release
    const void* kept = nullptr;
    void (*kept_drop)(const void*) = nullptr;
    ~LocalGroup() { if (kept && kept_drop) kept_drop(kept); }

    bool broken = false;   // a rank failed: every waiting rank leaves its collective with an error instead of hanging
    bool barrier(std::unique_lock<std::mutex>& lock) {
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != gen || broken; });
        }
        return !broken;
    }
    void account(uint32_t r) {  // (lock held) close rank r's current compute segment
        const auto now = std::chrono::steady_clock::now();
        if (!stage[r].empty()) {
            auto it = ms.find(stage[r]);
            if (it == ms.end()) {
                stage_order.push_back(stage[r]);
                it = ms.emplace(stage[r], std::vector<double>(world, 0.0)).first;
            }
            it->second[r] += std::chrono::duration<double, std::milli>(now - since[r]).count();
        }
        since[r] = now;
    }
    void acquire(uint32_t r, std::unique_lock<std::mutex>& lock) {
        cv.wait(lock, [&] { return turn == r || broken; });
        since[r] = std::chrono::steady_clock::now();
    }
    void release(uint32_t r) {  // (lock held)
        account(r);
        turn = r + 1;
        cv.notify_all();
    }
    // Lockstep runs put `world` ranks' working sets on ONE device (eight ranks of a 2^22-row proof: 21.8 GB of shared traces + 8 x
    // (20.4 GiB of tables + intermediates)).  A rank that waits in a collective gives its pool's cached blocks back to the driver
    // when the device runs short -- after its compute segment was closed, so the time is charged to no stage; a real multi-GPU
    // run has a device per rank and never gets here.
    static void trim_if_short(tvm_ctx* ctx) {
        size_t available = 0, total = 0;
        if (tvm_ctx_memory_info(ctx, &available, &total) == TVM_OK && available < total / 4) (void)tvm_ctx_trim(ctx);
    }
};

// side_slot < 0: the exchange on the context's stream, complete when the call returns.  side_slot >= 0 (tvmh_comm::all_gather_async):
// the same rendezvous, but the copies go on the context's SIDE LANE (include/triton_hip.h: tvm_side_*) behind the work queued on the
// context's stream so far and are marked in that slot -- the calls the RCCL communicator makes around ncclAllGather -- and the call
// returns with the copies in flight; local_wait completes it.
int32_t local_collective(void* self, tvm_ctx* ctx, const uint64_t* d_send, uint64_t* d_recv, uint64_t words, bool all_to_all, int side_slot = -1) {
    auto* mb = (LocalGroup::Member*)self;
    LocalGroup& g = *mb->group;
    const uint32_t r = mb->rank;
    int32_t status = tvm_sync(ctx);  // this rank's operands are complete (the peers read them on streams no event of this rank orders)
    std::unique_lock<std::mutex> lock(g.m);
    if (g.lockstep) {
        g.release(r);
        lock.unlock();
        LocalGroup::trim_if_short(ctx);
        lock.lock();
    }
    g.send[r] = d_send;
    if (!g.barrier(lock)) return TVM_ERR_DEVICE;
    std::vector<const uint64_t*> from = g.send;
    lock.unlock();
    if (side_slot >= 0 && status == TVM_OK) status = tvm_side_begin(ctx);
    for (uint32_t peer = 0; peer < g.world && status == TVM_OK; peer++) {
        uint64_t* dst = d_recv + (uint64_t)peer * words;
        const uint64_t* src = from[peer] + (all_to_all ? (uint64_t)r * words : 0);
        status = side_slot >= 0 ? tvm_side_memcpy_d2d(ctx, dst, src, words * 8) : tvm_memcpy_d2d(ctx, dst, src, words * 8);
    }
    if (status == TVM_OK) status = side_slot >= 0 ? tvm_side_mark(ctx, (uint32_t)side_slot) : tvm_sync(ctx);
    lock.lock();
    // synchronous: nobody reuses its send buffer before everybody has read it.  On the side lane the reads are still in flight: the
    // barrier only hands the turn on, local_wait's barrier is the one that protects the send buffers.
    if (!g.barrier(lock)) return TVM_ERR_DEVICE;
    if (g.lockstep) {
        if (r == 0) g.turn = 0, g.cv.notify_all();
        g.acquire(r, lock);
    }
    return status;
}
// tvmh_comm::wait: the context's stream waits for the slot's mark (its kernels may read the received words), the HOST for this rank's
// side lane (its copies out of the peers' send buffers are done), and a barrier tells every rank that its own send buffer has been
// read by everybody -- the caller may release it.
int32_t local_wait(void* self, tvm_ctx* ctx, uint32_t slot) {
    auto* mb = (LocalGroup::Member*)self;
    LocalGroup& g = *mb->group;
    const uint32_t r = mb->rank;
    int32_t status = tvm_side_wait(ctx, slot);
    const int32_t drained = tvm_side_sync(ctx);
    if (status == TVM_OK) status = drained;
    std::unique_lock<std::mutex> lock(g.m);
    if (g.lockstep) g.release(r);
    if (!g.barrier(lock)) return TVM_ERR_DEVICE;
    if (g.lockstep) {
        if (r == 0) g.turn = 0, g.cv.notify_all();
        g.acquire(r, lock);
    }
    return status;
}
// tvmh_comm::share (triton_host.hpp): a barrier (two, so that the lockstep turn passes as in a collective) around the hand-over
int32_t local_share(void* self, tvm_ctx* ctx, uint32_t op, const void* mine, void (*drop)(const void*), const void** out) {
    auto* mb = (LocalGroup::Member*)self;
    LocalGroup& g = *mb->group;
    const uint32_t r = mb->rank;
    int32_t status = tvm_sync(ctx);  // this rank's use of the kept object (release) / its work on the new one (publish) is complete
    std::unique_lock<std::mutex> lock(g.m);
    if (g.lockstep) {
        g.release(r);
        lock.unlock();
        LocalGroup::trim_if_short(ctx);
        lock.lock();
    }
    if (!g.barrier(lock)) return TVM_ERR_DEVICE;
    if (r == 0 && (op == TVMH_SHARE_RELEASE || (op == TVMH_SHARE_PUBLISH && mine))) {
        const void* old = g.kept;
        void (*old_drop)(const void*) = g.kept_drop;
        g.kept = op == TVMH_SHARE_PUBLISH ? mine : nullptr;
        g.kept_drop = op == TVMH_SHARE_PUBLISH ? drop : nullptr;
        if (out) *out = g.kept;  // from here on the group owns `mine`: the publisher learns it even if the barrier below fails
        if (old && old_drop) {   // rank 0's thread, rank 0's context: every rank has passed the barrier, nobody reads it any more
            lock.unlock();
            old_drop(old);
            lock.lock();
        }
    }
    if (!g.barrier(lock)) return TVM_ERR_DEVICE;
    if (out) *out = g.kept;
    if (g.lockstep) {
        if (r == 0) g.turn = 0, g.cv.notify_all();
        g.acquire(r, lock);
    }
    return status;
}
void local_abort(void* self) {
    LocalGroup& g = *((LocalGroup::Member*)self)->group;
    std::unique_lock<std::mutex> lock(g.m);
    g.broken = true;
    g.cv.notify_all();
}
int32_t local_all_gather(void* self, tvm_ctx* ctx, const uint64_t* s, uint64_t* d, uint64_t w) { return local_collective(self, ctx, s, d, w, false); }
int32_t local_all_to_all(void* self, tvm_ctx* ctx, const uint64_t* s, uint64_t* d, uint64_t w) { return local_collective(self, ctx, s, d, w, true); }
int32_t local_all_gather_async(void* self, tvm_ctx* ctx, const uint64_t* s, uint64_t* d, uint64_t w, uint32_t slot) {
    if (slot >= TVM_SIDE_SLOTS) return TVM_ERR_INVALID_ARGUMENT;
    return local_collective(self, ctx, s, d, w, false, (int)slot);
}
void local_begin(void* self, tvm_ctx*) {
    auto* mb = (LocalGroup::Member*)self;
    LocalGroup& g = *mb->group;
    if (!g.lockstep) return;
    std::unique_lock<std::mutex> lock(g.m);
    g.stage[mb->rank] = "setup";
    g.acquire(mb->rank, lock);
}
void local_mark(void* self, tvm_ctx* ctx, const char* name) {
    auto* mb = (LocalGroup::Member*)self;
    LocalGroup& g = *mb->group;
    if (!g.lockstep) return;
    (void)tvm_sync(ctx);
    std::unique_lock<std::mutex> lock(g.m);
    g.account(mb->rank);
    g.stage[mb->rank] = name;
}
void local_end(void* self, tvm_ctx* ctx) {
    auto* mb = (LocalGroup::Member*)self;
    LocalGroup& g = *mb->group;
    if (!g.lockstep) return;
    (void)tvm_sync(ctx);
    std::unique_lock<std::mutex> lock(g.m);
    g.release(mb->rank);
    g.stage[mb->rank].clear();
    if (mb->rank + 1 == g.world) g.turn = 0, g.cv.notify_all();  // the next proof starts with rank 0 again
}
}  // namespace

}  // namespace triton_vm

extern "C" int32_t tvmh_local_comms_create(uint32_t world, uint32_t lockstep, tvmh_comm** out) {
    using namespace triton_vm;
    if (!world || !out) return TVM_ERR_INVALID_ARGUMENT;
    auto* g = new (std::nothrow) LocalGroup();
    if (!g) return TVM_ERR_OUT_OF_MEMORY;
    g->world = world;
    g->lockstep = lockstep != 0;
    g->send.assign(world, nullptr);
    g->stage.assign(world, "");
    g->since.assign(world, std::chrono::steady_clock::now());
    g->members.resize(world);
    g->comms.resize(world);
    for (uint32_t r = 0; r < world; r++) {
        g->members[r] = {g, r};
        // The asynchronous exchange is offered by the FREE-RUNNING group only (the tests of the ordering logic).  A lockstep group is a
        // measurement of per-rank compute: its exchanges are real copies whose time is charged to no stage, and copies in flight on the
        // one shared GPU under a rank's kernels would be charged to them (4.9 GB per rank and proof: the column split measured 80 ms
        // that way against 48 with the exchange outside the turn, profiles/r06_e_*) -- so there the caller takes the synchronous path.
        g->comms[r] = tvmh_comm{&g->members[r], r, world, local_all_gather, local_all_to_all, local_begin, local_mark, local_end, local_abort, local_share,
                                lockstep ? nullptr : local_all_gather_async, lockstep ? nullptr : local_wait};
        out[r] = &g->comms[r];
    }
    return TVM_OK;
}
extern "C" void tvmh_local_comms_abort(tvmh_comm* any) {
    if (!any) return;
    triton_vm::local_abort(any->self);
}
extern "C" void tvmh_local_comms_destroy(tvmh_comm* first) {
    if (first) delete ((triton_vm::LocalGroup::Member*)first->self)->group;
}
extern "C" uint64_t tvmh_local_comms_report(const tvmh_comm* any, char* json, uint64_t capacity) {
    using namespace triton_vm;
    if (!any) return 0;
    LocalGroup& g = *((LocalGroup::Member*)any->self)->group;
    std::unique_lock<std::mutex> lock(g.m);
    std::string s = "{";
    char buf[48];
    for (size_t k = 0; k < g.stage_order.size(); k++) {
        s += std::string(k ? ", " : "") + "\"" + g.stage_order[k] + "\": [";
        const std::vector<double>& v = g.ms[g.stage_order[k]];
        for (size_t r = 0; r < v.size(); r++) {
            std::snprintf(buf, sizeof buf, "%s%.3f", r ? ", " : "", v[r]);
            s += buf;
        }
        s += "]";
    }
    s += "}";
    if (json && capacity > s.size()) std::memcpy(json, s.c_str(), s.size() + 1);
    return s.size() + 1;
}

namespace {
template <class F>
int32_t guarded(char* error, uint64_t error_capacity, F body) {
    using namespace triton_vm;
    try {
        body();
        return TVM_OK;
    } catch (const Error& e) {
        if (error && error_capacity) std::snprintf(error, error_capacity, "%s", e.what());
        return e.status ? e.status : TVM_ERR_INVALID_ARGUMENT;
    } catch (const std::exception& e) {
        if (error && error_capacity) std::snprintf(error, error_capacity, "%s", e.what());
        return TVM_ERR_DEVICE;
    }
}
}  // namespace

extern "C" int32_t tvmh_prove_execution_sharded(tvm_ctx* ctx, const tvmh_comm* comm, uint32_t jit_passes, uint64_t split_tree_min_leaves,
                                                const tvm_aet* aet, uint32_t log2_padded_height, uint32_t security_level,
                                                uint32_t log2_expansion, uint32_t use_stir, const uint8_t randomness_seed[32],
                                                const uint64_t* h_program_digest, const uint64_t* h_public_input, uint64_t n_public_input,
                                                const uint64_t* h_public_output, uint64_t n_public_output, uint64_t* h_proof,
                                                uint64_t capacity, uint64_t* proof_words, uint32_t profile, char* stats_json,
                                                uint64_t stats_capacity, char* error, uint64_t error_capacity) {
    using namespace triton_vm;
    return guarded(error, error_capacity, [&] {
        if (!aet || !randomness_seed) throw Error(TVM_ERR_INVALID_ARGUMENT, "tvmh_prove_execution_sharded: null execution trace or seed");
        if (use_stir > 2) throw Error(TVM_ERR_INVALID_ARGUMENT, "tvmh_prove_execution_sharded: use_stir is 0 (FRI), 1 (STIR) or 2 (automatic)");
        const Context c(ctx);
        const bool stir = use_stir == 2 ? log2_padded_height >= 16 : use_stir == 1;
        const StarkParameters p = stark_parameters(log2_padded_height, security_level, log2_expansion, stir);
        Claim claim;
        if (h_program_digest) std::memcpy(claim.program_digest, h_program_digest, sizeof(claim.program_digest));
        if (n_public_input) claim.input.assign(h_public_input, h_public_input + n_public_input);
        if (n_public_output) claim.output.assign(h_public_output, h_public_output + n_public_output);
        std::string stats;
        const std::vector<u64> proof = prove_execution_sharded(c, p, comm, jit_passes, *aet, claim, randomness_seed, profile != 0, &stats,
                                                               split_tree_min_leaves);
        if (stats_json && stats_capacity) std::snprintf(stats_json, stats_capacity, "%s", stats.c_str());
        if (proof_words) *proof_words = proof.size();
        if (h_proof && capacity >= proof.size()) std::memcpy(h_proof, proof.data(), proof.size() * sizeof(u64));
    });
}

extern "C" int32_t tvmh_prove_sharded(tvm_ctx* ctx, const tvmh_comm* comm, uint32_t jit_passes, uint64_t split_tree_min_leaves,
                                      uint32_t log2_padded_height, uint64_t num_trace_randomizers, uint64_t num_collinearity_checks,
                                      uint32_t log2_expansion, const uint64_t* d_main_trace, const uint64_t* d_main_randomizers,
                                      const uint64_t* d_aux_trace, const uint64_t* d_aux_randomizers, const uint64_t* h_quotient_randomizer,
                                      uint32_t use_stir, uint32_t stir_security_level, uint64_t* h_proof, uint64_t capacity,
                                      uint64_t* proof_words, char* error, uint64_t error_capacity) {
    using namespace triton_vm;
    return guarded(error, error_capacity, [&] {
        const Context c(ctx);
        StarkParameters p = use_stir ? stark_parameters(log2_padded_height, stir_security_level, log2_expansion, true)
                                     : StarkParameters(log2_padded_height, num_trace_randomizers, num_collinearity_checks, log2_expansion);
        if (use_stir && num_trace_randomizers) {
            // an explicitly sized instance (the tests' tiny tables): the caller's randomizer count with this STIR instance
            const Stir stir = p.stir;
            p = StarkParameters(log2_padded_height, num_trace_randomizers, num_collinearity_checks, log2_expansion);
            p.use_stir = true;
            p.stir = stir;
            p.ldt = stir.initial_domain;
        }
        std::vector<Xfe> qr(p.num_quotient_randomizers);
        std::memcpy(qr.data(), h_quotient_randomizer, qr.size() * sizeof(Xfe));
        const CommSession session(comm, c);
        ShardedProver prover(c, p, comm, jit_passes, d_main_trace, d_main_randomizers, d_aux_trace, d_aux_randomizers, qr);
        prover.split_tree_min_leaves = split_tree_min_leaves;
        const std::vector<u64> proof = prover.prove().proof();
        if (proof_words) *proof_words = proof.size();
        if (h_proof && capacity >= proof.size()) std::memcpy(h_proof, proof.data(), proof.size() * sizeof(u64));
    });
}
