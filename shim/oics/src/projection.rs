//! `oics::projection` (reference: packages/lib/src/projection.rs:17-194) -> omr_get_angle_with_projections.
use crate::bridge::view;
use crate::ffi;
use crate::transfer::TransformableMatrix;

/// Scale, gray, threshold, sweep -N..N candidates of `angle_step`, arg-max with the reference's tie policy.
/// `threads` is accepted and ignored: the reference's multi-thread branch rotates by the integer index
/// instead of index * step (projection.rs:94) and no caller uses it.  Panics where the reference panics
/// (`.expect`, projection.rs:26-31,57,61), because the signature has no error channel.
pub fn get_angle_with_projections(
    src: &TransformableMatrix,
    max_angle: u16,
    angle_step: f64,
    resize_scale: f64,
    threads: usize,
) -> f64 {
    let v = view(src.get_mat()).expect("8-bit image");
    let mut angle = 0.0f64;
    let rc = unsafe { ffi::omr_get_angle_with_projections(&v, max_angle, angle_step, resize_scale, threads, &mut angle) };
    if rc != ffi::OMR_OK {
        crate::bridge::check(rc).expect("get_angle_with_projections");
    }
    angle
}
