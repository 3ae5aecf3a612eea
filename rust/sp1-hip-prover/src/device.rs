//! Device, stream and buffer RAII around the runtime half of the C ABI (the role `sp1-gpu-cudart` plays for
//! `sp1-gpu-sys`, /root/reference/sp1-gpu/crates/cudart).
use core::ffi::c_void;
use std::ptr;

use sp1_hip_sys as sys;

use crate::error::{check, HipError};

/// One GPU of the node. One prover process per GPU is the deployment shape; the ordinal is what
/// `HIP_VISIBLE_DEVICES` leaves visible.
#[derive(Clone, Copy, Debug)]
pub struct HipDevice(pub i32);

impl HipDevice {
    /// Make this device current on the calling thread (every `sp1hip_*` call acts on the current device).
    pub fn set_current(self) -> Result<(), HipError> {
        check(unsafe { sys::sp1hip_set_device(self.0) })
    }
    pub fn count() -> Result<i32, HipError> {
        let mut n = 0;
        check(unsafe { sys::sp1hip_device_count(&mut n) })?;
        Ok(n)
    }
}

/// A HIP stream. The library is re-entrant per stream: one host thread per stream, any number of streams.
pub struct HipStream(sys::Stream);
// SAFETY: a hipStream_t may be used from any thread; the library keeps per-stream state behind its own locks.
unsafe impl Send for HipStream {}
unsafe impl Sync for HipStream {}

impl HipStream {
    pub fn new() -> Result<Self, HipError> {
        let mut s: sys::Stream = ptr::null_mut();
        check(unsafe { sys::sp1hip_stream_create(&mut s) })?;
        Ok(Self(s))
    }
    pub fn raw(&self) -> sys::Stream {
        self.0
    }
    pub fn synchronize(&self) -> Result<(), HipError> {
        check(unsafe { sys::sp1hip_stream_synchronize(self.0) })
    }
}

impl Drop for HipStream {
    fn drop(&mut self) {
        unsafe { sys::sp1hip_stream_destroy(self.0) };
    }
}

/// Pinned host words (`sp1hip_malloc_host`): the staging area host traces are generated into, so that
/// `sp1hip_stage_tables` copies at full PCIe rate. Counterpart of the reference's `PinnedBuffer`
/// (sp1-gpu/crates/cudart/src/pinned.rs) used by `host_main_tracegen`.
pub struct PinnedWords {
    ptr: *mut u32,
    len: usize,
}
unsafe impl Send for PinnedWords {}
unsafe impl Sync for PinnedWords {}

impl PinnedWords {
    pub fn with_capacity(len: usize) -> Result<Self, HipError> {
        let mut p: *mut c_void = ptr::null_mut();
        check(unsafe { sys::sp1hip_malloc_host(&mut p, len.max(1) * 4) })?;
        Ok(Self { ptr: p.cast(), len })
    }
    pub fn as_mut_slice(&mut self) -> &mut [u32] {
        // SAFETY: `ptr` is a live allocation of `len` words owned by self.
        unsafe { core::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
    pub fn as_ptr(&self) -> *const u32 {
        self.ptr
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
}

impl Drop for PinnedWords {
    fn drop(&mut self) {
        unsafe { sys::sp1hip_free_host(self.ptr.cast()) };
    }
}

/// One chip table on the device: COLUMN-major `[cols][rows]` Montgomery words (include/sp1hip.h, "Device layouts").
pub struct DeviceTable {
    ptr: *mut u32,
    pub rows: u64,
    pub cols: u32,
    stream: sys::Stream,
}
unsafe impl Send for DeviceTable {}
unsafe impl Sync for DeviceTable {}

impl DeviceTable {
    pub fn alloc(rows: u64, cols: u32, stream: &HipStream) -> Result<Self, HipError> {
        let mut p: *mut c_void = ptr::null_mut();
        let bytes = (rows as usize) * (cols as usize) * 4;
        check(unsafe { sys::sp1hip_malloc_async(&mut p, bytes.max(4), stream.raw()) })?;
        Ok(Self { ptr: p.cast(), rows, cols, stream: stream.raw() })
    }
    pub fn as_ptr(&self) -> *const u32 {
        self.ptr
    }
    pub fn as_mut_ptr(&mut self) -> *mut u32 {
        self.ptr
    }
    pub fn as_sys(&self) -> sys::Sp1HipTable {
        sys::Sp1HipTable { d_data: self.ptr, rows: self.rows, cols: self.cols }
    }
}

impl Drop for DeviceTable {
    fn drop(&mut self) {
        unsafe { sys::sp1hip_free_async(self.ptr.cast(), self.stream) };
    }
}

/// Host traces (row-major `[rows][cols]`, as `RowMajorMatrix` / `Mle` guts are) -> device tables, all chips of a shard
/// in ONE call: chunked H2D copies on a side stream overlap the on-GPU transposes (53 GB/s measured, PCIe-bound).
/// Replaces `device_main_tracegen`'s copy + `DeviceTensor::transpose` per chip
/// (/root/reference/sp1-gpu/crates/jagged_tracegen/src/lib.rs:L819-L835). The host slices must stay untouched until
/// `stream` has passed the call (pin them, or synchronise before dropping).
pub fn stage_tables(host: &[(&[u32], u64, u32)], stream: &HipStream) -> Result<Vec<DeviceTable>, HipError> {
    let mut out = Vec::with_capacity(host.len());
    for &(_, rows, cols) in host {
        out.push(DeviceTable::alloc(rows, cols, stream)?);
    }
    let descs: Vec<sys::Sp1HipHostTable> =
        host.iter().map(|&(words, rows, cols)| sys::Sp1HipHostTable { h_data: words.as_ptr(), rows, cols }).collect();
    let ptrs: Vec<*mut u32> = out.iter_mut().map(|t| t.as_mut_ptr()).collect();
    check(unsafe { sys::sp1hip_stage_tables(descs.as_ptr(), descs.len() as i32, ptrs.as_ptr(), stream.raw()) })?;
    Ok(out)
}
