//! `HipShardProver`: `AirProver` for a machine whose shards are proven by `libsp1hip.so` on one MI355X.
//!
//! Follows the CPU `ShardProver`'s implementation of the trait step by step
//! (/root/reference/crates/hypercube/src/prover/shard.rs:L245-L345): trace generation stays the reference's own
//! (`DefaultTraceGenerator`: the chips' `generate_trace` on the host — RISC-V trace generation needs the Rust executor's
//! records and is out of the backend's scope), then host traces -> device tables (`sp1hip_stage_tables`) and the whole
//! `prove_shard_with_data` (shard.rs:L650-L792) is ONE call, `sp1hip_prove_shard_with_pk`, whose bytes are
//! `bincode(ShardProof)`.
use std::{collections::BTreeMap, marker::PhantomData, sync::Arc};

use slop_algebra::PrimeField32;
use slop_alloc::CpuBackend;
use slop_challenger::IopCtx;
use sp1_hip_sys as sys;
use sp1_hypercube::{
    air::MachineAir,
    prover::{
        AirProver, DefaultTraceGenerator, MainTraceData, PcsProof, PreprocessedData, PreprocessedTraceData, Program,
        ProverPermit, ProverSemaphore, ProvingKey, Record, TraceData, TraceGenerator, Traces,
    },
    septic_digest::SepticDigest,
    Machine, MachineVerifyingKey, ShardContext, ShardProof, UntrustedConfig,
};

use crate::{
    device::{stage_tables, DeviceTable, HipDevice, HipStream},
    error::{check, HipError},
    machine::MachineDescription,
};

/// The shape parameters a `ShardVerifier` fixes (crates/prover/src/components.rs:L16-L17 for core:
/// stacking height 2^21, max_log_row_count 22; `core_fri_config()`: blow-up 4, 124 queries, 16 bits of grinding).
#[derive(Clone, Copy, Debug)]
pub struct ShardParams {
    pub max_log_row_count: i32,
    pub log_stacking_height: i32,
    /// Stacked columns per BaseFold batch: `interleave_multilinears_with_fixed_rate(32, ..)`
    /// (slop/crates/stacked/src/fixed_rate.rs:L6-L47).
    pub batch_size: i32,
    pub log_blowup: i32,
    pub num_queries: i32,
    pub proof_of_work_bits: i32,
}

impl ShardParams {
    fn as_sys(self) -> sys::Sp1HipShardParams {
        sys::Sp1HipShardParams {
            max_log_row_count: self.max_log_row_count,
            log_stacking_height: self.log_stacking_height,
            batch_size: self.batch_size,
            fri: sys::Sp1HipFriConfig {
                log_blowup: self.log_blowup,
                num_queries: self.num_queries,
                proof_of_work_bits: self.proof_of_work_bits,
            },
        }
    }
}

struct PkHandle(*mut sys::Sp1HipPk);
// SAFETY: the proving key is immutable after `sp1hip_setup`; provers on any number of streams read it concurrently
// (include/sp1hip.h, sp1hip_basefold_data_free: "a proving key shared by provers on their own streams").
unsafe impl Send for PkHandle {}
unsafe impl Sync for PkHandle {}
impl Drop for PkHandle {
    fn drop(&mut self) {
        unsafe { sys::sp1hip_pk_free(self.0) };
    }
}

/// `AirProver::PreprocessedData`: the device-side proving key (the preprocessed commitment round + the verifying key it
/// defines) and the preprocessed tables it was committed from, which every shard proof reads again.
pub struct HipProverData {
    pk: PkHandle,
    /// chip name -> device table, `BTreeMap` order = commitment order
    preprocessed: BTreeMap<String, DeviceTable>,
}

struct Inner<GC: IopCtx, SC: ShardContext<GC>> {
    trace_generator: DefaultTraceGenerator<GC::F, SC::Air, CpuBackend>,
    description: MachineDescription,
    params: ShardParams,
    device: HipDevice,
    _sc: PhantomData<SC>,
}

/// A shard prover on one GPU. `Arc`-cloned into `spawn_blocking` like the reference's.
pub struct HipShardProver<GC: IopCtx, SC: ShardContext<GC>> {
    inner: Arc<Inner<GC, SC>>,
}

impl<GC: IopCtx, SC: ShardContext<GC>> Clone for HipShardProver<GC, SC> {
    fn clone(&self) -> Self {
        Self { inner: self.inner.clone() }
    }
}

/// `&[F]` as the Montgomery words the library takes: `KoalaBear` is `#[repr(transparent)]` over its Montgomery `u32`
/// (the assumption sp1-gpu-sys makes, sys/src/dft.rs:L12-L59).
fn words<F: PrimeField32>(felts: &[F]) -> &[u32] {
    assert_eq!(core::mem::size_of::<F>(), 4);
    // SAFETY: F is a transparent wrapper of a u32 (asserted size; KoalaBear's layout).
    unsafe { core::slice::from_raw_parts(felts.as_ptr().cast(), felts.len()) }
}

impl<GC, SC> HipShardProver<GC, SC>
where
    GC: IopCtx,
    GC::F: PrimeField32,
    SC: ShardContext<GC>,
{
    /// `description`: the machine's chips as data — `sp1-hip-export --machine {core,compress,shrink}` writes it once.
    pub fn new(
        machine: Machine<GC::F, SC::Air>,
        description: MachineDescription,
        params: ShardParams,
        device: HipDevice,
    ) -> Result<Self, HipError> {
        for chip in machine.chips() {
            let d = description
                .chips
                .get(chip.name())
                .ok_or_else(|| HipError::Description(format!("no description for chip {}", chip.name())))?;
            if d.main_width as usize != chip.width() || d.prep_width as usize != chip.preprocessed_width() {
                return Err(HipError::Description(format!("chip {}: widths differ from the machine's", chip.name())));
            }
        }
        let trace_generator = DefaultTraceGenerator::new(machine);
        Ok(Self { inner: Arc::new(Inner { trace_generator, description, params, device, _sc: PhantomData }) })
    }

    fn max_log_row_count(&self) -> usize {
        self.inner.params.max_log_row_count as usize
    }

    /// Host traces of the chips (real rows only, row-major) -> column-major device tables, in `BTreeMap` (name) order.
    fn upload(&self, traces: &Traces<GC::F, CpuBackend>, stream: &HipStream) -> Result<BTreeMap<String, DeviceTable>, HipError> {
        let mut host = Vec::new();
        let mut names = Vec::new();
        for (name, trace) in traces.iter() {
            let rows = trace.num_real_entries() as u64;
            let cols = trace.num_polynomials() as u32;
            let slice: &[u32] = match trace.inner() {
                Some(mle) => words(mle.guts().as_slice()),
                None => &[],
            };
            host.push((slice, rows, cols));
            names.push(name.clone());
        }
        let tables = stage_tables(&host, stream)?;
        // pageable host memory: the copies are staged inside the call; pinned trace buffers (`PinnedWords`) make them
        // asynchronous, and then the caller keeps `traces` alive until the stream has passed
        stream.synchronize()?;
        Ok(names.into_iter().zip(tables).collect())
    }

    /// `setup_from_preprocessed_data_and_traces` (shard.rs:L406-L429): commit the preprocessed traces, get pk + vk.
    fn setup_blocking(
        &self,
        pc_start: [GC::F; 3],
        initial_global_cumulative_sum: SepticDigest<GC::F>,
        preprocessed_traces: &Traces<GC::F, CpuBackend>,
        untrusted_config: UntrustedConfig<GC::F>,
    ) -> Result<(HipProverData, MachineVerifyingKey<GC>), HipError>
    where
        GC::Digest: From<[GC::F; 8]>,
    {
        self.inner.device.set_current()?;
        let stream = HipStream::new()?;
        let preprocessed = self.upload(preprocessed_traces, &stream)?;
        let tables: Vec<sys::Sp1HipTable> = preprocessed.values().filter(|t| t.rows > 0).map(DeviceTable::as_sys).collect();
        let mut gcs = [0u32; 14];
        gcs[..7].copy_from_slice(words(&initial_global_cumulative_sum.0.x.0));
        gcs[7..].copy_from_slice(words(&initial_global_cumulative_sum.0.y.0));
        let mut pk: *mut sys::Sp1HipPk = core::ptr::null_mut();
        check(unsafe {
            sys::sp1hip_setup(
                tables.as_ptr(),
                tables.len() as i32,
                words(&pc_start).as_ptr(),
                gcs.as_ptr(),
                words(&[untrusted_config.enable_untrusted_programs])[0],
                self.inner.params.as_sys(),
                &mut pk,
                stream.raw(),
            )
        })?;
        let pk = PkHandle(pk);
        let mut vk_words = sys::Sp1HipVk {
            pc_start: [0; 3],
            initial_global_cumulative_sum: [0; 14],
            preprocessed_commit: [0; 8],
            enable_untrusted_programs: 0,
        };
        check(unsafe { sys::sp1hip_pk_vk(pk.0, &mut vk_words) })?;
        stream.synchronize()?;
        // SAFETY: Montgomery words -> field elements, the inverse of `words` above.
        let commit: [GC::F; 8] = unsafe { core::mem::transmute_copy(&vk_words.preprocessed_commit) };
        let vk = MachineVerifyingKey {
            pc_start,
            initial_global_cumulative_sum,
            preprocessed_commit: commit.into(),
            untrusted_config,
        };
        Ok((HipProverData { pk, preprocessed }, vk))
    }

    /// `prove_shard_with_data` (shard.rs:L650-L792) from the generated main traces on.
    fn prove_blocking(
        &self,
        pk: &HipProverData,
        main: MainTraceData<GC::F, SC::Air, CpuBackend>,
    ) -> Result<(ShardProof<GC, PcsProof<GC, SC>>, ProverPermit), HipError>
    where
        ShardProof<GC, PcsProof<GC, SC>>: serde::de::DeserializeOwned,
    {
        let MainTraceData { traces, public_values, shard_chips, permit } = main;
        self.inner.device.set_current()?;
        let stream = HipStream::new()?;
        let main_tables = self.upload(&traces, &stream)?;
        // one descriptor per chip of the shard's cluster, name order (`BTreeSet<Chip>`), absent chips with zero rows
        let mut chips = Vec::with_capacity(shard_chips.len());
        for chip in shard_chips.iter() {
            let name = chip.name();
            let d = &self.inner.description.chips[name];
            let m = &main_tables[name];
            let p = pk.preprocessed.get(name);
            chips.push(sys::Sp1HipShardChip {
                name: d.name.as_ptr(),
                program: d.constraints.as_ptr(),
                n_instr: (d.constraints.len() / 3) as u32,
                num_constraints: d.num_constraints,
                interactions: d.interactions.as_ptr(),
                n_words: d.interactions.len() as u32,
                main_width: d.main_width,
                prep_width: d.prep_width,
                d_main: if m.rows > 0 { m.as_ptr() } else { core::ptr::null() },
                d_prep: p.filter(|t| t.rows > 0).map_or(core::ptr::null(), |t| t.as_ptr()),
                real_rows: m.rows,
            });
        }
        let publics = words(&public_values);
        // size query, then the proof (SP1HIP_ERROR_BUFFER_TOO_SMALL sets the needed size and touches nothing else)
        let mut len = 0usize;
        let call = |buf: *mut u8, len: &mut usize| unsafe {
            sys::sp1hip_prove_shard_with_pk(
                pk.pk.0,
                chips.as_ptr(),
                chips.len() as i32,
                publics.as_ptr(),
                publics.len() as i32,
                core::ptr::null(),
                0,
                buf,
                len,
                stream.raw(),
            )
        };
        let status = call(core::ptr::null_mut(), &mut len);
        if status != sys::SP1HIP_ERROR_BUFFER_TOO_SMALL {
            check(status)?;
        }
        let mut bytes = vec![0u8; len];
        check(call(bytes.as_mut_ptr(), &mut len))?;
        bytes.truncate(len);
        stream.synchronize()?;
        let proof = bincode::deserialize(&bytes)?;
        Ok((proof, permit))
    }
}

impl<GC, SC> AirProver<GC, SC> for HipShardProver<GC, SC>
where
    GC: IopCtx,
    GC::F: PrimeField32,
    GC::Digest: From<[GC::F; 8]>,
    SC: ShardContext<GC>,
    ShardProof<GC, PcsProof<GC, SC>>: serde::de::DeserializeOwned,
{
    type PreprocessedData = HipProverData;

    fn machine(&self) -> &Machine<GC::F, SC::Air> {
        self.inner.trace_generator.machine()
    }

    async fn setup_from_vk(
        &self,
        program: Arc<Program<GC, SC>>,
        vk: Option<MachineVerifyingKey<GC>>,
        prover_permits: ProverSemaphore,
    ) -> (PreprocessedData<ProvingKey<GC, SC, Self>>, MachineVerifyingKey<GC>) {
        let initial_global_cumulative_sum = match vk {
            Some(vk) => vk.initial_global_cumulative_sum,
            None => {
                let program = program.clone();
                tokio::task::spawn_blocking(move || program.initial_global_cumulative_sum()).await.unwrap()
            }
        };
        let pc_start = program.pc_start();
        let untrusted_config = program.untrusted_config();
        let PreprocessedTraceData { preprocessed_traces, permit } = self
            .inner
            .trace_generator
            .generate_preprocessed_traces(program, self.max_log_row_count(), prover_permits)
            .await;
        let prover = self.clone();
        let (data, vk) = tokio::task::spawn_blocking(move || {
            prover.setup_blocking(pc_start, initial_global_cumulative_sum, &preprocessed_traces, untrusted_config)
        })
        .await
        .unwrap()
        .expect("sp1hip_setup failed");
        let pk = Arc::new(ProvingKey { vk: vk.clone(), preprocessed_data: data });
        (PreprocessedData { pk, permit }, vk)
    }

    async fn setup_and_prove_shard(
        &self,
        program: Arc<Program<GC, SC>>,
        record: Record<GC, SC>,
        vk: Option<MachineVerifyingKey<GC>>,
        prover_permits: ProverSemaphore,
    ) -> (MachineVerifyingKey<GC>, ShardProof<GC, PcsProof<GC, SC>>, ProverPermit) {
        let pc_start = program.pc_start();
        let untrusted_config = program.untrusted_config();
        let initial_global_cumulative_sum = match vk {
            Some(vk) => vk.initial_global_cumulative_sum,
            None => {
                let program = program.clone();
                tokio::task::spawn_blocking(move || program.initial_global_cumulative_sum()).await.unwrap()
            }
        };
        let TraceData { preprocessed_traces, main_trace_data } = self
            .inner
            .trace_generator
            .generate_traces(program, record, self.max_log_row_count(), prover_permits)
            .await;
        let prover = self.clone();
        tokio::task::spawn_blocking(move || {
            let (data, vk) = prover
                .setup_blocking(pc_start, initial_global_cumulative_sum, &preprocessed_traces, untrusted_config)
                .expect("sp1hip_setup failed");
            let (proof, permit) = prover.prove_blocking(&data, main_trace_data).expect("sp1hip_prove_shard_with_pk failed");
            (vk, proof, permit)
        })
        .await
        .unwrap()
    }

    async fn prove_shard_with_pk(
        &self,
        pk: Arc<ProvingKey<GC, SC, Self>>,
        record: Record<GC, SC>,
        prover_permits: ProverSemaphore,
    ) -> (ShardProof<GC, PcsProof<GC, SC>>, ProverPermit) {
        // (the library builds the default challenger and absorbs the verifying key itself: `vk.observe_into`)
        let main = self
            .inner
            .trace_generator
            .generate_main_traces(record, self.max_log_row_count(), prover_permits)
            .await;
        let prover = self.clone();
        // the calling thread spins on mapped pinned memory between sumcheck rounds: never on a tokio worker
        tokio::task::spawn_blocking(move || {
            prover.prove_blocking(&pk.preprocessed_data, main).expect("sp1hip_prove_shard_with_pk failed")
        })
        .await
        .unwrap()
    }

    async fn preprocessed_table_heights(pk: Arc<ProvingKey<GC, SC, Self>>) -> BTreeMap<String, usize> {
        pk.preprocessed_data.preprocessed.iter().map(|(name, t)| (name.clone(), t.rows as usize)).collect()
    }
}
