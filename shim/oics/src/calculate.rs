//! calculate.rs:2-23 of the reference through the ABI (sequential f64 mean, population std-dev).
use crate::ffi;

pub fn get_arithmetic_mean(vec: &Vec<f64>) -> f64 {
    let mut out = 0.0f64;
    let rc = unsafe { ffi::omr_get_arithmetic_mean(vec.as_ptr(), vec.len(), &mut out) };
    if rc != ffi::OMR_OK { f64::NAN } else { out } // an empty vector is 0/0 in the reference
}

pub fn get_standard_deviation(vec: &Vec<f64>) -> f64 {
    let mut out = 0.0f64;
    let rc = unsafe { ffi::omr_get_standard_deviation(vec.as_ptr(), vec.len(), &mut out) };
    if rc != ffi::OMR_OK { f64::NAN } else { out }
}
