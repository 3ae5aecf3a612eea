//! `SP1ProverComponents` for the MI355X backend — the plug-in point
//! (/root/reference/crates/prover/src/components.rs:L148-L173; the CUDA instance is
//! /root/reference/sp1-gpu/crates/prover_components/src/components.rs:L20-L27).
//!
//! Core and recursion (compress / shrink) shards run on the GPU. The wrap prover stays the reference's CPU prover: the
//! outer configuration is BN254-Poseidon2 (`SP1OuterGlobalContext`), which this KoalaBear backend does not implement
//! (the Groth16 / PLONK wrap is out of scope).
use sp1_core_machine::riscv::RiscvAir;
use sp1_hypercube::prover::{CpuShardProver, SP1OuterPcsProver};
use sp1_hypercube::SP1OuterPcs;
use sp1_primitives::{SP1Field, SP1GlobalContext, SP1OuterGlobalContext};
use sp1_prover::{CoreSC, RecursionSC, SP1ProverComponents, WrapAir, WrapProverBuilder};
use std::sync::Arc;

use crate::shard::HipShardProver;

pub struct SP1HipProverComponents;

/// The wrap prover is built on demand on the host, exactly like `CpuWrapProverBuilder` (components.rs:L177-L184).
pub struct HipWrapProverBuilder;

impl WrapProverBuilder<SP1HipProverComponents> for HipWrapProverBuilder {
    fn build(&self) -> Arc<<SP1HipProverComponents as SP1ProverComponents>::WrapProver> {
        let wrap_verifier = SP1HipProverComponents::wrap_verifier();
        Arc::new(CpuShardProver::new(wrap_verifier.shard_verifier().clone()))
    }
}

impl SP1ProverComponents for SP1HipProverComponents {
    /// `RiscvAir` shards (the BASELINE metric's path).
    type CoreProver = HipShardProver<SP1GlobalContext, CoreSC>;
    /// `CompressAir` shards: compress and shrink share the type, as in the reference (`with_shrink_air_prover` takes an
    /// `Arc<C::RecursionProver>`, crates/prover/src/worker/builder.rs:L150-L159).
    type RecursionProver = HipShardProver<SP1GlobalContext, RecursionSC>;
    type WrapProver = CpuShardProver<SP1OuterGlobalContext, SP1OuterPcs, SP1OuterPcsProver, WrapAir<SP1Field>>;
    type WrapProverBuilder = HipWrapProverBuilder;
}

#[allow(dead_code)]
fn _assert_core_air(_: &RiscvAir<SP1Field>) {}
