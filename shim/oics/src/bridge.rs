//! Glue between `opencv::core::Mat` and the C ABI's image views.  Nothing here computes.
use crate::ffi;
use opencv::core::{Mat, Scalar, CV_8U};
use opencv::prelude::*;
use std::ffi::CStr;
use std::os::raw::c_int;

/// Negative return code -> `opencv::Error` carrying the library's thread-local message
/// (the reference propagates `opencv::Error { code, message }` with `?`).
pub(crate) fn check(rc: c_int) -> opencv::Result<()> {
    if rc == ffi::OMR_OK {
        return Ok(());
    }
    let msg = unsafe {
        let p = ffi::omr_last_error();
        if p.is_null() { String::from("omrdeskew error") } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    Err(opencv::Error::new(rc, msg))
}

/// Borrow an 8-bit Mat as an `omr_image` (no copy; the Mat must outlive the call).
pub(crate) fn view(mat: &Mat) -> opencv::Result<ffi::OmrImage> {
    if mat.depth() != CV_8U {
        return Err(opencv::Error::new(ffi::OMR_ERR_ASSERT, String::from("8-bit images only")));
    }
    let step = mat.step1(0)? as i64; // elements per row == bytes per row for CV_8U
    Ok(ffi::OmrImage { data: mat.data(), rows: mat.rows(), cols: mat.cols(), channels: mat.channels(), step_bytes: step })
}

/// Copy a library-owned image into a fresh Mat and release the library's buffer.
pub(crate) fn into_mat(mut owned: ffi::OmrImageOwned) -> opencv::Result<Mat> {
    let typ = opencv::core::CV_MAKETYPE(CV_8U, owned.channels);
    let made = Mat::new_rows_cols_with_default(owned.rows, owned.cols, typ, Scalar::all(0.0));
    let result = made.and_then(|mut m| {
        let row_bytes = (owned.cols * owned.channels) as usize;
        for r in 0..owned.rows {
            unsafe {
                let src = owned.data.add(r as usize * owned.step_bytes as usize);
                std::ptr::copy_nonoverlapping(src, m.ptr_mut(r)?, row_bytes);
            }
        }
        Ok(m)
    });
    unsafe { ffi::omr_image_free(&mut owned) };
    result
}

/// `Scalar(b, g, r, a)` -> the four border bytes `omr_rotate` takes (saturate_cast<uchar>).
pub(crate) fn border_bytes(s: Scalar) -> [u8; 4] {
    let mut out = [0u8; 4];
    for k in 0..4 {
        out[k] = s[k].round().max(0.0).min(255.0) as u8;
    }
    out
}
