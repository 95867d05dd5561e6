//! `oics::hough` (reference: packages/lib/src/hough.rs:17-100) -> omr_get_angle_with_hough.
use crate::bridge::{check, into_mat, view};
use crate::ffi;
use crate::transfer::TransformableMatrix;
use opencv::core::Vector;
use opencv::imgcodecs;
use std::path::Path;

/// Canny(50, 150) -> HoughLinesP(1, pi/180, 0, min_line_length, max_line_gap) -> f32 atan2, `% 45`,
/// mode within 0.1 deg, all on the GPU.  The reference also writes a debug picture of the edge map into
/// `edge_image_output_dir` (hough.rs:46-63,94-99); that is kept, as host-side codec work: the edge map comes
/// from omr_canny and is written when the directory is not empty.
pub fn get_angle_with_hough(
    gray_tm: &TransformableMatrix,
    min_line_length: f64,
    max_line_gap: f64,
    file_name: &str,
    edge_image_output_dir: &str,
) -> Result<f64, opencv::Error> {
    let v = view(gray_tm.get_mat())?;
    let mut angle = 0.0f64;
    check(unsafe { ffi::omr_get_angle_with_hough(&v, min_line_length, max_line_gap, &mut angle) })?;
    if !edge_image_output_dir.is_empty() {
        let mut edges = ffi::OmrImageOwned::empty();
        check(unsafe { ffi::omr_canny(&v, 50.0, 150.0, &mut edges) })?;
        let pic = into_mat(edges)?;
        let path = Path::new(edge_image_output_dir).join(file_name);
        let params: Vector<i32> = Vector::from_slice(&[imgcodecs::IMWRITE_JPEG_QUALITY, 100]);
        imgcodecs::imwrite(path.to_str().unwrap_or(file_name), &pic, &params)?;
    }
    Ok(angle)
}
