//! `oics` on the MI355X: the public surface of the reference crate (packages/lib/src/lib.rs:1-12)
//! over libomrdeskew.so.  Callers keep their `use oics::{core, imgcodecs, imgproc, ...}` lines.
pub use opencv::{
    core, highgui, imgcodecs, imgproc, prelude, types as opencv_types, Result as OpenCV_Result,
};

pub mod calculate;
pub mod constants;
pub mod fft;
pub mod hough;
pub mod omr;
pub mod projection;
pub mod transfer;
pub mod types;

mod bridge;
pub mod ffi;
