//! `ProverPool`: N shard proofs in flight on one GPU — a safe wrapper of `sp1hip_pool_*`.
//!
//! The reference bounds concurrent proofs with a `ProverSemaphore` (crates/hypercube/src/prover/permits.rs:L36-L66); its
//! GPU builder takes ONE permit (sp1-gpu/crates/prover_components/src/builder.rs:L107) because its prover owns the device.
//! Here the library owns the concurrency: `n_slots` prover slots (thread + stream each) fill each other's transcript
//! hand-over gaps and a stager thread uploads the next shards' host traces meanwhile. A worker that wants the pool instead
//! of one `spawn_blocking` per shard submits from `prove_shard_with_pk` and awaits the ticket.
use std::ptr;

use sp1_hip_sys as sys;

use crate::error::{check, HipError};

pub struct ProverPool {
    raw: *mut sys::Sp1HipPool,
}
// SAFETY: the pool is internally synchronised (a mutex around its queues); submit / wait may be called from any thread.
unsafe impl Send for ProverPool {}
unsafe impl Sync for ProverPool {}

/// One submitted shard. Everything the chips point to must outlive `wait` — keep the owners next to the ticket.
#[must_use = "a ticket must be waited for: the pool keeps the proof until then"]
pub struct PoolTicket(pub sys::Ticket);

impl ProverPool {
    pub fn new(device: i32, n_slots: usize) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::sp1hip_pool_create(device, n_slots as i32, &mut raw) })?;
        Ok(Self { raw })
    }

    /// # Safety
    /// `pk` and every pointer inside `chips` must stay valid and unchanged until the ticket has been waited for.
    pub unsafe fn submit(
        &self,
        pk: *const sys::Sp1HipPk,
        chips: &[sys::Sp1HipPoolChip],
        public_values: &[u32],
    ) -> Result<PoolTicket, HipError> {
        let mut t: sys::Ticket = 0;
        check(sys::sp1hip_pool_submit(
            self.raw,
            pk,
            chips.as_ptr(),
            chips.len() as i32,
            if public_values.is_empty() { ptr::null() } else { public_values.as_ptr() },
            public_values.len() as i32,
            &mut t,
        ))?;
        Ok(PoolTicket(t))
    }

    /// Blocks (call from `spawn_blocking`): `bincode(ShardProof)` and where the ticket spent its time.
    pub fn wait(&self, ticket: PoolTicket) -> Result<(Vec<u8>, sys::Sp1HipPoolTimes), HipError> {
        let mut len = 0usize;
        let mut times = sys::Sp1HipPoolTimes { staging_ms: 0.0, queued_ms: 0.0, proving_ms: 0.0, slot: -1 };
        let st = unsafe { sys::sp1hip_pool_wait(self.raw, ticket.0, ptr::null_mut(), &mut len, &mut times) };
        if st != sys::SP1HIP_ERROR_BUFFER_TOO_SMALL {
            check(st)?;
        }
        let mut bytes = vec![0u8; len];
        check(unsafe { sys::sp1hip_pool_wait(self.raw, ticket.0, bytes.as_mut_ptr(), &mut len, &mut times) })?;
        bytes.truncate(len);
        Ok((bytes, times))
    }
}

impl Drop for ProverPool {
    fn drop(&mut self) {
        // finishes every submitted shard first
        unsafe { sys::sp1hip_pool_destroy(self.raw) };
    }
}
