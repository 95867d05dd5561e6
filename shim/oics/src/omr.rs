//! `oics::omr` (reference: packages/lib/src/omr.rs) over the C ABI.
use crate::bridge::{check, into_mat, view};
use crate::ffi;
use opencv::core::{Mat, Vector};
use opencv::imgcodecs;
use opencv::prelude::*;

/// omr.rs:41-45
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ResultStatus {
    Believed,
    NeedCheck,
    NotAResult,
}

/// omr.rs:46-50
#[derive(Clone, Debug)]
pub struct OmrResult {
    pub angle: f64,
    pub status: ResultStatus,
    pub candidates: Vec<f64>,
}

fn status_of(code: i32) -> ResultStatus {
    match code {
        ffi::OMR_STATUS_BELIEVED => ResultStatus::Believed,
        ffi::OMR_STATUS_NEED_CHECK => ResultStatus::NeedCheck,
        _ => ResultStatus::NotAResult,
    }
}

/// Runs `call` with a growing candidate buffer until every candidate fits.
fn with_candidates<F>(mut call: F) -> opencv::Result<OmrResult>
where
    F: FnMut(*mut f64, *mut i32, *mut f64, i32, *mut i32) -> std::os::raw::c_int,
{
    let mut cap: i32 = 1024;
    loop {
        let mut cand = vec![0.0f64; cap as usize];
        let (mut angle, mut status, mut len) = (0.0f64, 0i32, 0i32);
        check(call(&mut angle, &mut status, cand.as_mut_ptr(), cap, &mut len))?;
        if len <= cap {
            cand.truncate(len as usize);
            return Ok(OmrResult { angle, status: status_of(status), candidates: cand });
        }
        cap = len;
    }
}

/// omr.rs:8-39 -> (horizontal projection [rows], vertical projection [cols]).
pub fn get_mat_projection_data(mat: &Mat) -> opencv::Result<(Vec<f64>, Vec<f64>)> {
    let v = view(mat)?;
    let mut h = vec![0.0f64; v.rows as usize];
    let mut w = vec![0.0f64; v.cols as usize];
    check(unsafe { ffi::omr_get_mat_projection_data(&v, h.as_mut_ptr(), w.as_mut_ptr()) })?;
    Ok((h, w))
}

/// omr.rs:52-229.
pub fn get_result_from_projection(
    src: &Mat,
    projection_max_angle: u16,
    projection_angle_step: f64,
    projection_max_width: i32,
    projection_max_height: i32,
) -> opencv::Result<OmrResult> {
    let v = view(src)?;
    with_candidates(|a, s, c, cap, n| unsafe {
        ffi::omr_get_result_from_projection(&v, projection_max_angle, projection_angle_step, projection_max_width, projection_max_height, a, s, c, cap, n)
    })
}

/// omr.rs:231-302.
pub fn get_result_from_edges_detection(
    src: &Mat,
    edges_min_line_length: f64,
    edges_max_line_gap: f64,
) -> opencv::Result<OmrResult> {
    let v = view(src)?;
    with_candidates(|a, s, c, cap, n| unsafe {
        ffi::omr_get_result_from_edges_detection(&v, edges_min_line_length, edges_max_line_gap, a, s, c, cap, n)
    })
}

/// omr.rs:304-337.
pub fn get_result_from_fourier_transform(
    src: &Mat,
    canny_threshold_weak: f64,
    canny_threshold_strong: f64,
    fourier_min_line_length: f64,
    fourier_max_line_gap: f64,
) -> opencv::Result<OmrResult> {
    let v = view(src)?;
    with_candidates(|a, s, c, cap, n| unsafe {
        ffi::omr_get_result_from_fourier_transform(&v, canny_threshold_weak, canny_threshold_strong, fourier_min_line_length, fourier_max_line_gap, a, s, c, cap, n)
    })
}

/// omr.rs:339-448: imread(COLOR) -> projection, Hough fallback and the decision on the GPU -> CONTAIN warp
/// on the GPU -> imwrite(JPEG, quality 100).  Returns (rotation angle, need_check).
pub fn correct_default(
    input_file: &str,
    output_file: &str,
    projection_max_angle: u16,
    projection_angle_step: f64,
    projection_max_width: i32,
    projection_max_height: i32,
    hough_min_line_length: f64,
    hough_max_line_gap: f64,
) -> opencv::Result<(f64, bool)> {
    let src = imgcodecs::imread(input_file, imgcodecs::IMREAD_COLOR)?;
    let (mut angle, mut need_check) = (0.0f64, 0i32);
    let mut rotated = ffi::OmrImageOwned::empty();
    check(unsafe {
        ffi::omr_correct_default(&view(&src)?, projection_max_angle, projection_angle_step, projection_max_width,
                                 projection_max_height, hough_min_line_length, hough_max_line_gap, &mut angle,
                                 &mut need_check, &mut rotated)
    })?;
    let out = into_mat(rotated)?;
    let params: Vector<i32> = Vector::from_slice(&[imgcodecs::IMWRITE_JPEG_QUALITY, 100]);
    imgcodecs::imwrite(output_file, &out, &params)?;
    Ok((angle, need_check != 0))
}
