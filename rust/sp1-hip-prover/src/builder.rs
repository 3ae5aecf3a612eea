//! Worker start-up: `hip_worker_builder` is `cuda_worker_builder`
//! (/root/reference/sp1-gpu/crates/prover_components/src/builder.rs:L95-L134) for this backend. One worker process per
//! GPU (`HIP_VISIBLE_DEVICES=k`), as `sp1-gpu-server` runs one per device id; shards are independent, so a node scales by
//! running eight of these — no collective anywhere on the proving path.
use std::sync::Arc;

use sp1_core_executor::SP1CoreOpts;
use sp1_core_machine::riscv::RiscvAir;
use sp1_hypercube::{prover::ProverSemaphore, Machine};
use sp1_primitives::{
    fri_params::{core_fri_config, recursion_fri_config, shrink_fri_config},
    SP1Field, SP1GlobalContext,
};
use sp1_prover::{
    worker::SP1WorkerBuilder, CompressAir, CoreSC, RecursionSC, CORE_LOG_STACKING_HEIGHT, CORE_MAX_LOG_ROW_COUNT,
    SHRINK_LOG_STACKING_HEIGHT, SHRINK_MAX_LOG_ROW_COUNT,
};

use crate::{
    components::{HipWrapProverBuilder, SP1HipProverComponents},
    device::HipDevice,
    error::HipError,
    machine::MachineDescription,
    shard::{HipShardProver, ShardParams},
};

/// Shard proofs in flight per GPU. The reference's GPU builder uses ONE permit (builder.rs:L107): its prover owns the
/// device. This library is re-entrant per stream and 20-25 % of a single proof is GPU idle time behind transcript
/// hand-overs, so a second and third prover fill those gaps.
pub const HIP_PROVER_PERMITS: usize = 3;

fn params(fri: slop_primitives::FriConfig<SP1Field>, log_stacking_height: u32, max_log_row_count: usize) -> ShardParams {
    ShardParams {
        max_log_row_count: max_log_row_count as i32,
        log_stacking_height: log_stacking_height as i32,
        batch_size: 32,
        log_blowup: fri.log_blowup as i32,
        num_queries: fri.num_queries as i32,
        proof_of_work_bits: fri.proof_of_work_bits as i32,
    }
}

/// The core prover. `core_description`: output of `sp1-hip-export --machine core`.
pub fn hip_core_prover(
    machine: Machine<SP1Field, RiscvAir<SP1Field>>,
    core_description: &str,
    device: HipDevice,
) -> Result<HipShardProver<SP1GlobalContext, CoreSC>, HipError> {
    HipShardProver::new(
        machine,
        MachineDescription::from_json(core_description)?,
        params(core_fri_config(), CORE_LOG_STACKING_HEIGHT, CORE_MAX_LOG_ROW_COUNT),
        device,
    )
}

/// The compress (`shrink == false`) or shrink prover. `description`: `sp1-hip-export --machine compress | shrink`.
pub fn hip_recursion_prover(
    description: &str,
    shrink: bool,
    device: HipDevice,
) -> Result<HipShardProver<SP1GlobalContext, RecursionSC>, HipError> {
    use sp1_verifier::compressed::{RECURSION_LOG_STACKING_HEIGHT, RECURSION_MAX_LOG_ROW_COUNT};
    let (machine, p) = if shrink {
        (
            CompressAir::<SP1Field>::shrink_machine(),
            params(shrink_fri_config(), SHRINK_LOG_STACKING_HEIGHT, SHRINK_MAX_LOG_ROW_COUNT),
        )
    } else {
        (
            CompressAir::<SP1Field>::compress_machine(),
            params(recursion_fri_config(), RECURSION_LOG_STACKING_HEIGHT, RECURSION_MAX_LOG_ROW_COUNT),
        )
    };
    HipShardProver::new(machine, MachineDescription::from_json(description)?, p, device)
}

/// `SP1WorkerBuilder` with the GPU provers installed (`with_core_air_prover` / `with_compress_air_prover` /
/// `with_shrink_air_prover`, crates/prover/src/worker/builder.rs:L128-L190). `descriptions`: the three JSON documents.
pub fn hip_worker_builder(
    device: HipDevice,
    core_description: &str,
    compress_description: &str,
    shrink_description: &str,
) -> Result<SP1WorkerBuilder<SP1HipProverComponents>, HipError> {
    device.set_current()?;
    let machine = RiscvAir::<SP1Field>::machine();
    let permits = ProverSemaphore::new(HIP_PROVER_PERMITS);
    let core = Arc::new(hip_core_prover(machine.clone(), core_description, device)?);
    let compress = Arc::new(hip_recursion_prover(compress_description, false, device)?);
    let shrink = Arc::new(hip_recursion_prover(shrink_description, true, device)?);
    // 288 GB of HBM: full-size shards, nothing dropped or recomputed (the reference trims both for 24 GB cards,
    // sp1-gpu/crates/prover_components/src/builder.rs:L27-L60)
    let mut opts = SP1CoreOpts::default();
    opts.shard_size = 1 << 24;
    opts.global_dependencies_opt = true;
    Ok(SP1WorkerBuilder::new_with_machine(machine)
        .with_core_opts(opts)
        .with_core_air_prover(core, permits.clone())
        .with_compress_air_prover(compress, permits.clone())
        .with_shrink_air_prover(shrink, permits.clone())
        .with_wrap_air_prover(HipWrapProverBuilder, permits))
}
