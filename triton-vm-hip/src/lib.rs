//! Safe wrappers over `libtriton_hip.so` for the seams of `triton-vm`'s prover (feature `hip`, see
//! `patches/triton-vm-hip.patch` and INTEGRATION.md).  SOURCE ONLY: the image this was written in has no Rust
//! toolchain, so this crate has never been compiled; `ffi.rs` is generated from `include/triton_hip.h` and checked
//! for completeness by `tests/test_rust_shim.py`.
//!
//! Everything Fiat–Shamir (`ProofStream`, `BFieldCodec`, `sample_scalars`, `sample_indices`), the prover's RNG and
//! `MerkleTree::authentication_structure` stay in Rust/twenty-first; only plain words cross the boundary:
//! `BFieldElement::raw_u64()` Montgomery words, `XFieldElement` = 3 words, `Digest` = 5 words.
pub mod ffi;

use std::cell::RefCell;
use std::ffi::{c_void, CStr};
use std::ptr;

use ndarray::{Array2, ArrayView2};
use twenty_first::prelude::*;

use ffi::*;

#[derive(Debug, Clone, PartialEq, Eq)]
pub enum HipError {
    /// `TVM_ERR_INVALID_ARGUMENT`: domain / length mismatches (ArithmeticDomainError and friends)
    InvalidArgument(String),
    /// `TVM_ERR_OUT_OF_MEMORY`: the reference's `try_reserve_exact` failure (master_table.rs:268-271); recoverable
    OutOfMemory,
    /// `TVM_ERR_DEVICE` / `TVM_ERR_UNSUPPORTED`
    Device(String),
}
pub type Result<T> = std::result::Result<T, HipError>;

/// One context per proving thread (`triton_vm::prove` may run on several threads, lib.rs:522-532).
pub struct Context {
    raw: *mut TvmCtx,
}

thread_local! {
    static CONTEXT: RefCell<Option<Context>> = const { RefCell::new(None) };
}

/// Run `f` with this thread's context (created on first use on device `TRITON_HIP_DEVICE`, default 0).
pub fn with_context<T>(f: impl FnOnce(&Context) -> Result<T>) -> Result<T> {
    CONTEXT.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.is_none() {
            let device = std::env::var("TRITON_HIP_DEVICE").ok().and_then(|d| d.parse().ok()).unwrap_or(0);
            *slot = Some(Context::new(device)?);
        }
        f(slot.as_ref().unwrap())
    })
}

impl Context {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        let status = unsafe { tvm_ctx_create(device, ptr::null_mut(), &mut raw) };
        if status != TVM_OK {
            let what = unsafe { CStr::from_ptr(tvm_status_string(status)) }.to_string_lossy().into_owned();
            return Err(HipError::Device(what));
        }
        Ok(Self { raw })
    }

    fn check(&self, status: i32) -> Result<()> {
        match status {
            TVM_OK => Ok(()),
            TVM_ERR_OUT_OF_MEMORY => Err(HipError::OutOfMemory),
            TVM_ERR_INVALID_ARGUMENT => Err(HipError::InvalidArgument(self.last_error())),
            _ => Err(HipError::Device(self.last_error())),
        }
    }

    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(tvm_last_error(self.raw)) }.to_string_lossy().into_owned()
    }

    pub fn alloc(&self, words: usize) -> Result<DeviceBuffer<'_>> {
        let mut p: *mut c_void = ptr::null_mut();
        self.check(unsafe { tvm_malloc(self.raw, words.max(1) * 8, &mut p) })?;
        Ok(DeviceBuffer { ctx: self, ptr: p.cast(), words })
    }

    pub fn upload(&self, words: &[u64]) -> Result<DeviceBuffer<'_>> {
        let buf = self.alloc(words.len())?;
        self.check(unsafe { tvm_memcpy_h2d(self.raw, buf.ptr.cast(), words.as_ptr().cast(), words.len() * 8) })?;
        Ok(buf)
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { tvm_ctx_destroy(self.raw) }
    }
}

/// Device memory from the context's caching allocator.
pub struct DeviceBuffer<'c> {
    ctx: &'c Context,
    ptr: *mut u64,
    words: usize,
}

impl DeviceBuffer<'_> {
    pub fn download(&self) -> Result<Vec<u64>> {
        let mut out = vec![0_u64; self.words];
        self.ctx.check(unsafe { tvm_memcpy_d2h(self.ctx.raw, out.as_mut_ptr().cast(), self.ptr.cast(), self.words * 8) })?;
        Ok(out)
    }
}

impl Drop for DeviceBuffer<'_> {
    fn drop(&mut self) {
        unsafe { tvm_free(self.ctx.raw, self.ptr.cast()) };
    }
}

/// A low-degree-extended master table resident on the device (`tvm_table`).
pub struct Table<'c> {
    ctx: &'c Context,
    raw: *mut TvmTable,
}

impl Drop for Table<'_> {
    fn drop(&mut self) {
        unsafe { tvm_table_free(self.ctx.raw, self.raw) }
    }
}

// ---- words <-> field elements -------------------------------------------------------------------------------------
pub fn domain(offset: BFieldElement, generator: BFieldElement, length: usize) -> TvmDomain {
    TvmDomain { offset: offset.raw_u64(), generator: generator.raw_u64(), length: length as u64 }
}

pub fn bfe_words(elements: &[BFieldElement]) -> Vec<u64> {
    elements.iter().map(|e| e.raw_u64()).collect()
}

pub fn xfe_words(elements: &[XFieldElement]) -> Vec<u64> {
    elements.iter().flat_map(|x| x.coefficients.map(|c| c.raw_u64())).collect()
}

pub fn words_to_xfes(words: &[u64]) -> Vec<XFieldElement> {
    words
        .chunks_exact(3)
        .map(|w| XFieldElement::new([0, 1, 2].map(|i| BFieldElement::from_raw_u64(w[i]))))
        .collect()
}

pub fn words_to_digests(words: &[u64]) -> Vec<Digest> {
    words
        .chunks_exact(Digest::LEN)
        .map(|w| Digest::new([0, 1, 2, 3, 4].map(|i| BFieldElement::from_raw_u64(w[i]))))
        .collect()
}

/// Column-major words of a trace table of `field_kind`-word elements (`Array2::f()` order, master_table.rs:888,1013).
fn column_major_words<F: Copy>(table: ArrayView2<F>, field_kind: usize, words_of: impl Fn(F, &mut Vec<u64>)) -> Vec<u64> {
    let mut out = Vec::with_capacity(table.len() * field_kind);
    for column in table.columns() {
        for &cell in column {
            words_of(cell, &mut out);
        }
    }
    out
}

// ---- seams ---------------------------------------------------------------------------------------------------------
/// `MasterTable::maybe_low_degree_extend_all_columns` (master_table.rs:258-322) for a main table.
/// `randomizers[c]` = the coefficients of `trace_randomizer_for_column(c)` (host RNG, master_table.rs:423-434).
pub fn lde_main_table<'c>(
    ctx: &'c Context,
    trace: ArrayView2<BFieldElement>,
    randomizers: &[Vec<BFieldElement>],
    trace_domain: TvmDomain,
    evaluation_domain: TvmDomain,
) -> Result<Table<'c>> {
    let h = randomizers.first().map_or(0, Vec::len);
    let d_trace = ctx.upload(&column_major_words(trace, 1, |c: BFieldElement, out| out.push(c.raw_u64())))?;
    let d_rand = ctx.upload(&randomizers.iter().flat_map(|r| bfe_words(r)).collect::<Vec<_>>())?;
    lde(ctx, 1, &d_trace, trace.nrows(), trace.ncols(), &d_rand, h, trace_domain, evaluation_domain)
}

/// The same for the auxiliary table (XFieldElement cells, 3 words each).
pub fn lde_aux_table<'c>(
    ctx: &'c Context,
    trace: ArrayView2<XFieldElement>,
    randomizers: &[Vec<XFieldElement>],
    trace_domain: TvmDomain,
    evaluation_domain: TvmDomain,
) -> Result<Table<'c>> {
    let h = randomizers.first().map_or(0, Vec::len);
    let words = column_major_words(trace, 3, |c: XFieldElement, out| out.extend(c.coefficients.map(|b| b.raw_u64())));
    let d_trace = ctx.upload(&words)?;
    let d_rand = ctx.upload(&randomizers.iter().flat_map(|r| xfe_words(r)).collect::<Vec<_>>())?;
    lde(ctx, 3, &d_trace, trace.nrows(), trace.ncols(), &d_rand, h, trace_domain, evaluation_domain)
}

#[allow(clippy::too_many_arguments)]
fn lde<'c>(
    ctx: &'c Context,
    field_kind: i32,
    d_trace: &DeviceBuffer,
    n_rows: usize,
    n_cols: usize,
    d_rand: &DeviceBuffer,
    h: usize,
    trace_domain: TvmDomain,
    evaluation_domain: TvmDomain,
) -> Result<Table<'c>> {
    let mut raw = ptr::null_mut();
    ctx.check(unsafe {
        tvm_lde_table(ctx.raw, field_kind, d_trace.ptr, n_rows as u64, n_cols as u64, d_rand.ptr, h as u64, trace_domain,
                      evaluation_domain, &mut raw)
    })?;
    Ok(Table { ctx, raw })
}

impl Table<'_> {
    pub fn num_rows(&self) -> usize {
        unsafe { tvm_table_num_rows(self.raw) as usize }
    }

    /// The reference's row-major `Array2` (master_table.rs:304-305) -- the acceptance-test path that feeds the
    /// unmodified downstream code; the production path keeps the table on the device.
    pub fn to_host_bfe(&self) -> Result<Array2<BFieldElement>> {
        let (rows, cols) = (self.num_rows(), unsafe { tvm_table_num_columns(self.raw) as usize });
        let buf = self.ctx.alloc(rows * cols)?;
        self.ctx.check(unsafe { tvm_table_export_row_major(self.ctx.raw, self.raw, buf.ptr) })?;
        let words = buf.download()?;
        Ok(Array2::from_shape_vec([rows, cols], words.into_iter().map(BFieldElement::from_raw_u64).collect()).unwrap())
    }

    pub fn to_host_xfe(&self) -> Result<Array2<XFieldElement>> {
        let (rows, cols) = (self.num_rows(), unsafe { tvm_table_num_columns(self.raw) as usize });
        let buf = self.ctx.alloc(rows * cols * 3)?;
        self.ctx.check(unsafe { tvm_table_export_row_major(self.ctx.raw, self.raw, buf.ptr) })?;
        Ok(Array2::from_shape_vec([rows, cols], words_to_xfes(&buf.download()?)).unwrap())
    }

    /// `MasterTable::hash_all_ldt_domain_rows` (master_table.rs:455-468)
    pub fn hash_all_ldt_domain_rows(&self, ldt_length: usize) -> Result<Vec<Digest>> {
        let buf = self.ctx.alloc(ldt_length * Digest::LEN)?;
        self.ctx.check(unsafe { tvm_hash_rows(self.ctx.raw, self.raw, ldt_length as u64, buf.ptr) })?;
        Ok(words_to_digests(&buf.download()?))
    }

    /// `MasterTable::reveal_rows` (master_table.rs:548-555, cached branch): `[indices.len()][row words]`
    pub fn reveal_rows(&self, ldt_length: usize, indices: &[usize]) -> Result<Vec<u64>> {
        let idx: Vec<u64> = indices.iter().map(|&i| i as u64).collect();
        let row_words = unsafe { (tvm_table_num_columns(self.raw) as usize) * (tvm_table_field_kind(self.raw) as usize) };
        let mut out = vec![0_u64; idx.len() * row_words];
        self.ctx.check(unsafe {
            tvm_table_reveal_rows(self.ctx.raw, self.raw, ldt_length as u64, idx.as_ptr(), idx.len() as u64, out.as_mut_ptr())
        })?;
        Ok(out)
    }
}

/// `all_quotients_combined` (master_table.rs:1264-1363) on two device tables: 63 challenges, 604 quotient weights.
pub fn all_quotients_combined(
    ctx: &Context,
    main: &Table,
    aux: &Table,
    trace_domain: TvmDomain,
    quotient_domain: TvmDomain,
    challenges: &[XFieldElement],
    quotient_weights: &[XFieldElement],
) -> Result<Vec<XFieldElement>> {
    assert_eq!(TVM_NUM_CHALLENGES, challenges.len());
    assert_eq!(TVM_NUM_QUOTIENT_WEIGHTS, quotient_weights.len());
    let out = ctx.alloc(quotient_domain.length as usize * 3)?;
    let (ch, w) = (xfe_words(challenges), xfe_words(quotient_weights));
    ctx.check(unsafe {
        tvm_all_quotients_combined(ctx.raw, main.raw, aux.raw, trace_domain, quotient_domain, ch.as_ptr(), w.as_ptr(), out.ptr)
    })?;
    Ok(words_to_xfes(&out.download()?))
}
pub const TVM_NUM_CHALLENGES: usize = 63;
pub const TVM_NUM_QUOTIENT_WEIGHTS: usize = 604;

/// `ProverRound::split_and_fold` (low_degree_test/fri.rs:349-366)
pub fn fri_split_and_fold(ctx: &Context, codeword: &[XFieldElement], fri_domain: TvmDomain, challenge: XFieldElement) -> Result<Vec<XFieldElement>> {
    let d_in = ctx.upload(&xfe_words(codeword))?;
    let d_out = ctx.alloc(codeword.len() / 2 * 3)?;
    let ch = xfe_words(&[challenge]);
    ctx.check(unsafe { tvm_fri_split_and_fold(ctx.raw, d_in.ptr, fri_domain, ch.as_ptr(), d_out.ptr) })?;
    Ok(words_to_xfes(&d_out.download()?))
}

/// All nodes of the Merkle tree over `Digest::from(xfe)` leaves (`merkle_tree_from_codeword`, fri.rs:343-347), heap
/// order `[2n][5]` (node 1 = root, leaves at n..2n), for cross-checking twenty-first's `MerkleTree::par_new`.
pub fn codeword_merkle_nodes(ctx: &Context, codeword: &[XFieldElement]) -> Result<Vec<Digest>> {
    let d_in = ctx.upload(&xfe_words(codeword))?;
    let d_nodes = ctx.alloc(2 * codeword.len() * Digest::LEN)?;
    ctx.check(unsafe { tvm_codeword_merkle_tree(ctx.raw, d_in.ptr, codeword.len() as u64, d_nodes.ptr) })?;
    Ok(words_to_digests(&d_nodes.download()?))
}

/// `MasterMainTable::extend`'s per-table loops + the degree-lowering fill (master_table.rs:1006-1075) on the device:
/// `aux` is `[n_rows, 91]` column-major words with column 90 (batch randomizer) already drawn by the host RNG.
pub fn extend_aux_table(ctx: &Context, main_trace_words: &[u64], aux_trace_words: &mut [u64], n_rows: usize, challenges: &[XFieldElement]) -> Result<()> {
    let d_main = ctx.upload(main_trace_words)?;
    let d_aux = ctx.upload(aux_trace_words)?;
    assert_eq!(TVM_NUM_CHALLENGES, challenges.len());   // the C side reads 63 * 3 words and n_rows-sized columns unconditionally
    assert_eq!(379 * n_rows, main_trace_words.len());
    assert_eq!(91 * n_rows * 3, aux_trace_words.len());
    let ch = xfe_words(challenges);
    ctx.check(unsafe { tvm_extend_aux_table(ctx.raw, d_main.ptr, d_aux.ptr, n_rows as u64, ch.as_ptr()) })?;
    ctx.check(unsafe { tvm_fill_derived_aux_columns(ctx.raw, d_main.ptr, d_aux.ptr, n_rows as u64, ch.as_ptr()) })?;
    aux_trace_words.copy_from_slice(&d_aux.download()?);
    Ok(())
}

// ---- stage 2: the whole of Prover::prove(claim, aet) behind one call ---------------------------------------------
/// The algebraic execution trace as plain words, field by field as `AlgebraicExecutionTrace` holds it (aet.rs:41-96):
/// trace arrays row-major in `raw_u64` Montgomery words, multiplicities as plain integers.  `u32_entries`:
/// `[opcode, left operand, right operand, multiplicity]` per entry in `IndexMap` order; `cascade_entries`:
/// `[16-bit limb, multiplicity]`.
pub struct ExecutionTrace {
    pub program_words: Vec<u64>,
    pub instruction_multiplicities: Vec<u32>,
    pub processor_trace: Vec<u64>,
    pub op_stack_trace: Vec<u64>,
    pub ram_trace: Vec<u64>,
    pub program_hash_trace: Vec<u64>,
    pub sponge_trace: Vec<u64>,
    pub hash_trace: Vec<u64>,
    pub u32_entries: Vec<u64>,
    pub cascade_entries: Vec<u64>,
    pub lookup_multiplicities: Vec<u64>,
}

/// `Stark::ldt_choice`: `Auto` = the reference's heuristic (STIR from 2^16 padded rows on, stark.rs:1944-1951).
#[derive(Debug, Copy, Clone, PartialEq, Eq)]
pub enum Ldt {
    Fri = 0,
    Stir = 1,
    Auto = 2,
}

/// `tvm_aet` of include/triton_hip.h (the arrays may be host or device memory; here: host).
#[repr(C)]
struct TvmAet {
    program_words: *const u64,
    instruction_multiplicities: *const u32,
    program_len: u64,
    processor_trace: *const u64,
    processor_len: u64,
    op_stack_trace: *const u64,
    op_stack_len: u64,
    ram_trace: *const u64,
    ram_len: u64,
    bezout_coefficients_0: *const u64,
    bezout_coefficients_1: *const u64,
    num_ram_pointers: u64,
    program_hash_trace: *const u64,
    program_hash_len: u64,
    sponge_trace: *const u64,
    sponge_len: u64,
    hash_trace: *const u64,
    hash_len: u64,
    u32_entries: *const u64,
    u32_len: u64,
    cascade_entries: *const u64,
    cascade_len: u64,
    lookup_multiplicities: *const u64,
}

// libtriton_host.so (triton_vm_amd/host/triton_host.hpp): the C++ mirror of Prover::prove above the C ABI
#[link(name = "triton_host")]
unsafe extern "C" {
    fn tvmh_prove_execution(
        ctx: *mut TvmCtx,
        aet: *const TvmAet,
        log2_padded_height: u32,
        security_level: u32,
        log2_expansion: u32,
        use_stir: u32,
        randomness_seed: *const u8,
        h_program_digest: *const u64,
        h_public_input: *const u64,
        n_public_input: u64,
        h_public_output: *const u64,
        n_public_output: u64,
        h_proof: *mut u64,
        capacity: u64,
        proof_words: *mut u64,
        error: *mut std::ffi::c_char,
        error_capacity: u64,
    ) -> i32;
}

/// the `tvm_aet` view of an `ExecutionTrace` (lengths checked; the Bezout coefficient polynomials are left to the device)
fn raw_aet(aet: &ExecutionTrace) -> TvmAet {
    assert_eq!(aet.program_words.len(), aet.instruction_multiplicities.len());
    assert_eq!(256, aet.lookup_multiplicities.len());
    for (words, width) in [
        (&aet.processor_trace, 39),
        (&aet.op_stack_trace, 4),
        (&aet.ram_trace, 7),
        (&aet.program_hash_trace, 67),
        (&aet.sponge_trace, 67),
        (&aet.hash_trace, 67),
        (&aet.u32_entries, 4),
        (&aet.cascade_entries, 2),
    ] {
        assert_eq!(0, words.len() % width);
    }
    TvmAet {
        program_words: aet.program_words.as_ptr(),
        instruction_multiplicities: aet.instruction_multiplicities.as_ptr(),
        program_len: aet.program_words.len() as u64,
        processor_trace: aet.processor_trace.as_ptr(),
        processor_len: (aet.processor_trace.len() / 39) as u64,
        op_stack_trace: aet.op_stack_trace.as_ptr(),
        op_stack_len: (aet.op_stack_trace.len() / 4) as u64,
        ram_trace: aet.ram_trace.as_ptr(),
        ram_len: (aet.ram_trace.len() / 7) as u64,
        bezout_coefficients_0: ptr::null(),   // both null: computed on the device from the sorted RAM table
        bezout_coefficients_1: ptr::null(),
        num_ram_pointers: 0,
        program_hash_trace: aet.program_hash_trace.as_ptr(),
        program_hash_len: (aet.program_hash_trace.len() / 67) as u64,
        sponge_trace: aet.sponge_trace.as_ptr(),
        sponge_len: (aet.sponge_trace.len() / 67) as u64,
        hash_trace: aet.hash_trace.as_ptr(),
        hash_len: (aet.hash_trace.len() / 67) as u64,
        u32_entries: aet.u32_entries.as_ptr(),
        u32_len: (aet.u32_entries.len() / 4) as u64,
        cascade_entries: aet.cascade_entries.as_ptr(),
        cascade_len: (aet.cascade_entries.len() / 2) as u64,
        lookup_multiplicities: aet.lookup_multiplicities.as_ptr(),
    }
}

/// `Prover::prove(claim, aet)` (stark.rs:331-719) on the device: fill, pad, randomizers, extend, LDE, Merkle trees, AIR,
/// quotient segments, DEEP, the low-degree test and the openings with the master tables resident in HBM throughout;
/// the RAM table's Bezout coefficient polynomials are computed on the device.  Returns the `raw_u64` words of the
/// reference's `Proof` -- for the same seed, the words the CPU prover emits.
#[allow(clippy::too_many_arguments)]
pub fn prove_execution(
    ctx: &Context,
    aet: &ExecutionTrace,
    padded_height: usize,
    security_level: usize,
    log2_expansion: usize,
    ldt: Ldt,
    randomness_seed: &[u8; 32],
    program_digest: &[u64],
    public_input: &[u64],
    public_output: &[u64],
) -> Result<Vec<u64>> {
    // Acceptance switch of run_acceptance.sh, compiled in ONLY under the cargo feature `acceptance-sharded-entry` (off by default: a
    // prover behind triton_vm::prove() must not change code paths because of an inherited environment -- the libraries read none, and
    // neither does this crate as shipped).  With the feature, TRITON_HIP_SHARDED_ENTRY=<passes> routes the same call through
    // `tvmh_prove_execution_sharded` with no communicator -- the entry point of the multi-GPU / coset-wise prover under the
    // reference's memory policy (0 = the policy decides) -- so that the reference's own tests check that code path too; a value that
    // does not parse is an error, not a silent 0.
    #[cfg(feature = "acceptance-sharded-entry")]
    if let Some(passes) = std::env::var_os("TRITON_HIP_SHARDED_ENTRY") {
        let jit_passes = passes
            .to_str()
            .and_then(|p| p.parse::<u32>().ok())
            .unwrap_or_else(|| panic!("TRITON_HIP_SHARDED_ENTRY must be a pass count (0 = the memory policy), got {passes:?}"));
        return prove_execution_sharded(
            ctx, None, jit_passes, aet, padded_height, security_level, log2_expansion, ldt, randomness_seed, program_digest, public_input,
            public_output,
        );
    }
    assert!(padded_height.is_power_of_two());
    assert_eq!(5, program_digest.len());
    let raw = raw_aet(aet);
    let mut proof = vec![0_u64; 1 << 20];
    let mut error = [0 as std::ffi::c_char; 512];
    loop {
        let mut n = 0_u64;
        let status = unsafe {
            tvmh_prove_execution(
                ctx.raw,
                &raw,
                padded_height.trailing_zeros(),
                security_level as u32,
                log2_expansion as u32,
                ldt as u32,
                randomness_seed.as_ptr(),
                program_digest.as_ptr(),
                public_input.as_ptr(),
                public_input.len() as u64,
                public_output.as_ptr(),
                public_output.len() as u64,
                proof.as_mut_ptr(),
                proof.len() as u64,
                &mut n,
                error.as_mut_ptr(),
                error.len() as u64,
            )
        };
        match status {
            TVM_OK if n as usize <= proof.len() => {
                proof.truncate(n as usize);
                return Ok(proof);
            }
            TVM_OK => proof = vec![0_u64; n as usize],   // the proof did not fit: grow and run again
            TVM_ERR_OUT_OF_MEMORY => return Err(HipError::OutOfMemory),
            TVM_ERR_INVALID_ARGUMENT => {
                return Err(HipError::InvalidArgument(unsafe { CStr::from_ptr(error.as_ptr()) }.to_string_lossy().into_owned()))
            }
            _ => return Err(HipError::Device(unsafe { CStr::from_ptr(error.as_ptr()) }.to_string_lossy().into_owned())),
        }
    }
}

// ---- one proof over the GPUs of a node (triton_vm_amd/host/sharded_host.cpp, rccl_comm.cpp) ----------------
/// `tvmh_comm` of triton_host.hpp: the collectives of the sharded prover as a table of functions.  `libtriton_rccl.so` fills it
/// with RCCL calls on the context's stream; a Rust deployment only passes the pointer along.
#[repr(C)]
pub struct TvmhComm {
    _private: [u8; 0],
}

/// Anything that owns a `tvmh_comm` (with the `rccl` feature: [`RcclComm`]).
pub trait Communicator {
    fn raw(&self) -> *const TvmhComm;
}

#[cfg(feature = "rccl")]
#[link(name = "triton_rccl")]
unsafe extern "C" {
    /// rank 0 draws the 128-byte `ncclUniqueId`; the launcher carries it to the other ranks
    fn tvmh_rccl_unique_id(out: *mut u8) -> i32;
    fn tvmh_rccl_comm_create(unique_id: *const u8, rank: u32, world: u32, device: i32, out: *mut *mut TvmhComm) -> i32;
    fn tvmh_rccl_comm_destroy(comm: *mut TvmhComm);
}

#[link(name = "triton_host")]
unsafe extern "C" {
    fn tvmh_prove_execution_sharded(
        ctx: *mut TvmCtx,
        comm: *const TvmhComm,
        jit_passes: u32,
        split_tree_min_leaves: u64,
        aet: *const TvmAet,
        log2_padded_height: u32,
        security_level: u32,
        log2_expansion: u32,
        use_stir: u32,
        randomness_seed: *const u8,
        h_program_digest: *const u64,
        h_public_input: *const u64,
        n_public_input: u64,
        h_public_output: *const u64,
        n_public_output: u64,
        h_proof: *mut u64,
        capacity: u64,
        proof_words: *mut u64,
        profile: u32,
        stats_json: *mut std::ffi::c_char,
        stats_capacity: u64,
        error: *mut std::ffi::c_char,
        error_capacity: u64,
    ) -> i32;
}

/// One rank's RCCL communicator (one process per GPU).  Feature `rccl`.
#[cfg(feature = "rccl")]
pub struct RcclComm {
    raw: *mut TvmhComm,
}

#[cfg(feature = "rccl")]
impl Communicator for RcclComm {
    fn raw(&self) -> *const TvmhComm {
        self.raw as *const TvmhComm
    }
}

#[cfg(feature = "rccl")]
impl RcclComm {
    pub fn unique_id() -> Result<[u8; 128]> {
        let mut id = [0_u8; 128];
        match unsafe { tvmh_rccl_unique_id(id.as_mut_ptr()) } {
            TVM_OK => Ok(id),
            _ => Err(HipError::Device("ncclGetUniqueId failed".into())),
        }
    }

    pub fn new(unique_id: &[u8; 128], rank: u32, world: u32, device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        match unsafe { tvmh_rccl_comm_create(unique_id.as_ptr(), rank, world, device, &mut raw) } {
            TVM_OK => Ok(Self { raw }),
            _ => Err(HipError::Device("ncclCommInitRank failed".into())),
        }
    }
}

#[cfg(feature = "rccl")]
impl Drop for RcclComm {
    fn drop(&mut self) {
        unsafe { tvmh_rccl_comm_destroy(self.raw) }
    }
}

/// `prove_execution` over the ranks of `comm` (every rank passes the same trace, claim and seed and obtains the same proof),
/// and / or coset by coset: `jit_passes` = 0 is the reference's memory policy (cached first, master_table.rs:268-271).
#[allow(clippy::too_many_arguments)]
pub fn prove_execution_sharded(
    ctx: &Context,
    comm: Option<&dyn Communicator>,
    jit_passes: u32,
    aet: &ExecutionTrace,
    padded_height: usize,
    security_level: usize,
    log2_expansion: usize,
    ldt: Ldt,
    randomness_seed: &[u8; 32],
    program_digest: &[u64],
    public_input: &[u64],
    public_output: &[u64],
) -> Result<Vec<u64>> {
    assert!(padded_height.is_power_of_two());
    assert_eq!(5, program_digest.len());
    let raw = raw_aet(aet);
    let mut proof = vec![0_u64; 1 << 20];
    let mut error = [0 as std::ffi::c_char; 512];
    loop {
        let mut n = 0_u64;
        let status = unsafe {
            tvmh_prove_execution_sharded(
                ctx.raw,
                comm.map_or(ptr::null(), |c| c.raw()),
                jit_passes,
                1 << 21,
                &raw,
                padded_height.trailing_zeros(),
                security_level as u32,
                log2_expansion as u32,
                ldt as u32,
                randomness_seed.as_ptr(),
                program_digest.as_ptr(),
                public_input.as_ptr(),
                public_input.len() as u64,
                public_output.as_ptr(),
                public_output.len() as u64,
                proof.as_mut_ptr(),
                proof.len() as u64,
                &mut n,
                0,
                ptr::null_mut(),
                0,
                error.as_mut_ptr(),
                error.len() as u64,
            )
        };
        match status {
            TVM_OK if n as usize <= proof.len() => {
                proof.truncate(n as usize);
                return Ok(proof);
            }
            TVM_OK => proof = vec![0_u64; n as usize],
            TVM_ERR_OUT_OF_MEMORY => return Err(HipError::OutOfMemory),
            TVM_ERR_INVALID_ARGUMENT => {
                return Err(HipError::InvalidArgument(unsafe { CStr::from_ptr(error.as_ptr()) }.to_string_lossy().into_owned()))
            }
            _ => return Err(HipError::Device(unsafe { CStr::from_ptr(error.as_ptr()) }.to_string_lossy().into_owned())),
        }
    }
}
