use core::ffi::c_int;

/// A failed `sp1hip_*` call: the status code of include/sp1hip.h and the library's thread-local message.
#[derive(Debug, thiserror::Error)]
pub enum HipError {
    #[error("sp1hip status {status}: {message}")]
    Status { status: c_int, message: String },
    #[error("machine description: {0}")]
    Description(String),
    #[error("proof bytes do not deserialize: {0}")]
    Decode(#[from] bincode::Error),
}

pub(crate) fn check(status: c_int) -> Result<(), HipError> {
    sp1_hip_sys::check(status).map_err(|(status, message)| HipError::Status { status, message })
}
