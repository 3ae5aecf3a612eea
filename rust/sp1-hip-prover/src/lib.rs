//! `sp1-hip-prover`: SP1's `AirProver` seam on an MI355X through `libsp1hip.so`.
//!
//! UNCOMPILED in the repository that produced it (no Rust toolchain in its image); written against succinctlabs/sp1
//! v6.4.0. What it mirrors in the reference:
//!
//! * `HipShardProver`      — `CudaShardProver` as an `AirProver` (crates/hypercube/src/prover/shard.rs:L45-L109 is the
//!                            trait; the CPU implementation it follows step by step is shard.rs:L245-L345)
//! * `SP1HipProverComponents` — `SP1CudaProverComponents` (sp1-gpu/crates/prover_components/src/components.rs:L20-L27)
//! * `hip_worker_builder`  — `cuda_worker_builder` (sp1-gpu/crates/prover_components/src/builder.rs:L95-L134)
//! * `ProverPool`          — N shard proofs in flight per GPU: the library-side counterpart of `ProverSemaphore`
//!                            (crates/hypercube/src/prover/permits.rs:L36-L66)
//!
//! The whole proof is ONE FFI call (`sp1hip_prove_shard_with_pk`): commit -> LogUp-GKR -> zerocheck -> jagged evaluation
//! proof, returning `bincode(ShardProof)`. The chips' constraints and lookups travel as data (`MachineDescription`, the JSON
//! of `sp1-hip-export`).
mod builder;
mod components;
mod device;
mod error;
mod machine;
mod pool;
mod shard;

pub use builder::{hip_core_prover, hip_recursion_prover, hip_worker_builder, HIP_PROVER_PERMITS};
pub use components::SP1HipProverComponents;
pub use device::{DeviceTable, HipDevice, HipStream, PinnedWords};
pub use error::HipError;
pub use machine::{ChipProgram, MachineDescription};
pub use pool::{PoolTicket, ProverPool};
pub use shard::{HipProverData, HipShardProver, ShardParams};
