//! `oics::transfer` (reference: packages/lib/src/transfer.rs) over the C ABI.
//! File decode / encode / windows stay host-side OpenCV calls; every pixel operation the corrector's
//! paths use (gray, threshold, resize, rotate, projections) runs on the GPU.
use crate::bridge::{border_bytes, check, into_mat, view};
use crate::ffi;
use crate::types::{ImageFormat, RotateClipStrategy};
use opencv::core::{Mat, Scalar, Vector};
use opencv::prelude::*;
use opencv::{highgui, imgcodecs, imgproc};

/// Owning wrapper over a `Mat` (transfer.rs:16-18).
pub struct TransformableMatrix {
    matrix: Mat,
}

// the reference asserts this too (transfer.rs:279); the wrapped Mat is only read concurrently
unsafe impl Sync for TransformableMatrix {}

impl TransformableMatrix {
    pub fn default() -> Self {
        TransformableMatrix { matrix: Mat::default() }
    }

    pub fn load_mat(self: &mut Self, filename: &str, flags: i32) -> Result<&mut Self, opencv::Error> {
        self.matrix = imgcodecs::imread(filename, flags)?;
        Ok(self)
    }

    pub fn get_mat(self: &Self) -> &Mat {
        &self.matrix
    }

    pub fn from_matrix(mat: &Mat) -> Self {
        TransformableMatrix { matrix: mat.clone() }
    }

    pub fn new(filename: &str, flags: i32) -> Result<Self, opencv::Error> {
        Ok(TransformableMatrix { matrix: imgcodecs::imread(filename, flags)? })
    }

    /// transfer.rs:66-91 -> omr_scale (INTER_LINEAR when enlarging, INTER_AREA otherwise).
    pub fn scale_self(self: &mut Self, scale: f64) -> Result<&mut Self, opencv::Error> {
        if scale == 1.0 {
            return Ok(self);
        }
        let mut out = ffi::OmrImageOwned::empty();
        check(unsafe { ffi::omr_scale(&view(&self.matrix)?, scale, &mut out) })?;
        self.matrix = into_mat(out)?;
        Ok(self)
    }

    /// transfer.rs:93-126 -> omr_shrink_to (never enlarges).
    pub fn shrink_to(self: &mut Self, max_width: i32, max_height: i32) -> Result<&mut Self, opencv::Error> {
        let mut out = ffi::OmrImageOwned::empty();
        check(unsafe { ffi::omr_shrink_to(&view(&self.matrix)?, max_width, max_height, &mut out) })?;
        self.matrix = into_mat(out)?;
        Ok(self)
    }

    /// transfer.rs:128-145 -> omr_resize (INTER_AREA).
    pub fn resize_self(self: &mut Self, width: i32, height: i32) -> Result<&mut Self, opencv::Error> {
        let mut out = ffi::OmrImageOwned::empty();
        check(unsafe { ffi::omr_resize(&view(&self.matrix)?, width, height, &mut out) })?;
        self.matrix = into_mat(out)?;
        Ok(self)
    }

    pub fn show(self: &Self, win_name: &str) -> Result<(), opencv::Error> {
        highgui::named_window(win_name, highgui::WINDOW_NORMAL)?;
        highgui::imshow(win_name, &self.matrix)
    }

    pub fn get_bytes(self: &Self) -> Result<&[u8], opencv::Error> {
        self.matrix.data_bytes()
    }

    pub fn im_write(self: &Self, filename: &str, format: ImageFormat, quality: i32) -> Result<bool, opencv::Error> {
        let key = match format {
            ImageFormat::JPEG => imgcodecs::IMWRITE_JPEG_QUALITY,
            ImageFormat::PNG => imgcodecs::IMWRITE_PNG_COMPRESSION,
            ImageFormat::WEBP => imgcodecs::IMWRITE_WEBP_QUALITY,
        };
        let params: Vector<i32> = Vector::from_slice(&[key, quality]);
        imgcodecs::imwrite(filename, &self.matrix, &params)
    }

    pub fn clone(&self) -> Self {
        TransformableMatrix { matrix: self.matrix.clone() }
    }

    /// Visualisation helper outside the corrector's paths (transfer.rs:206-231): host OpenCV.
    pub fn dilate(&self, kernel_shape: i32, kernel_size: opencv::core::Size, anchor: opencv::core::Point, iterations: i32) -> opencv::Result<Self> {
        let kernel = imgproc::get_structuring_element(kernel_shape, kernel_size, anchor)?;
        let mut dst = Mat::default();
        imgproc::dilate(&self.matrix, &mut dst, &kernel, anchor, iterations, opencv::core::BORDER_CONSTANT, imgproc::morphology_default_border_value()?)?;
        Ok(TransformableMatrix { matrix: dst })
    }

    /// Visualisation helper outside the corrector's paths (transfer.rs:254-277): host OpenCV.
    /// (The erode that IS on path 2, omr.rs:98-112, runs inside omr_get_result_from_projection.)
    pub fn erode(&self, kernel_shape: i32, kernel_size: opencv::core::Size, anchor: opencv::core::Point, iterations: i32) -> opencv::Result<Self> {
        let kernel = imgproc::get_structuring_element(kernel_shape, kernel_size, anchor)?;
        let mut dst = Mat::default();
        imgproc::erode(&self.matrix, &mut dst, &kernel, anchor, iterations, opencv::core::BORDER_CONSTANT, imgproc::morphology_default_border_value()?)?;
        Ok(TransformableMatrix { matrix: dst })
    }
}

fn new_u8c1(rows: i32, cols: i32) -> opencv::Result<Mat> {
    Mat::new_rows_cols_with_default(rows, cols, opencv::core::CV_8UC1, Scalar::all(0.0))
}

/// transfer.rs:283-290: cvtColor(RGB2GRAY) -> omr_rgb_to_gray.
pub fn transfer_rgb_image_to_gray_image(src: &TransformableMatrix) -> Result<TransformableMatrix, opencv::Error> {
    let v = view(&src.matrix)?;
    let mut dst = new_u8c1(v.rows, v.cols)?;
    let step = dst.step1(0)? as i64;
    check(unsafe { ffi::omr_rgb_to_gray(&v, dst.data_mut(), step) })?;
    Ok(TransformableMatrix { matrix: dst })
}

/// transfer.rs:294-301: threshold(127, 255, BINARY) -> omr_threshold_binary.
pub fn transfer_gray_image_to_thresh_binary(src: &TransformableMatrix) -> Result<TransformableMatrix, opencv::Error> {
    let v = view(&src.matrix)?;
    let mut dst = new_u8c1(v.rows, v.cols)?;
    let step = dst.step1(0)? as i64;
    check(unsafe { ffi::omr_threshold_binary(&v, dst.data_mut(), step) })?;
    Ok(TransformableMatrix { matrix: dst })
}

/// transfer.rs:305-333: black pixels per row.
pub fn get_horizontal_projection(src: &TransformableMatrix) -> Result<Vec<f64>, opencv::Error> {
    let v = view(&src.matrix)?;
    let mut out = vec![0.0f64; v.rows as usize];
    check(unsafe { ffi::omr_get_horizontal_projection(&v, out.as_mut_ptr()) })?;
    Ok(out)
}

/// transfer.rs:380-405: black pixels per column.
pub fn get_vertical_projection(src: &TransformableMatrix) -> Result<Vec<f64>, opencv::Error> {
    let v = view(&src.matrix)?;
    let mut out = vec![0.0f64; v.cols as usize];
    check(unsafe { ffi::omr_get_vertical_projection(&v, out.as_mut_ptr()) })?;
    Ok(out)
}

/// Debug picture (transfer.rs:337-376): row r gets its black count as a bar from the left edge.
pub fn transfer_thresh_binary_to_horizontal_projection(src: &TransformableMatrix) -> Result<TransformableMatrix, opencv::Error> {
    let counts = get_horizontal_projection(src)?;
    let (rows, cols) = (src.matrix.rows(), src.matrix.cols());
    let mut pic = Mat::new_rows_cols_with_default(rows, cols, opencv::core::CV_8UC1, Scalar::all(255.0))?;
    for r in 0..rows {
        let n = (counts[r as usize] as i32).min(cols);
        let row = pic.at_row_mut::<u8>(r)?;
        for px in row.iter_mut().take(n as usize) {
            *px = 0;
        }
    }
    Ok(TransformableMatrix { matrix: pic })
}

/// Debug picture (transfer.rs:409-455): column c gets its black count as a bar from the bottom edge.
pub fn transfer_thresh_binary_to_vertical_projection(src: &TransformableMatrix) -> Result<TransformableMatrix, opencv::Error> {
    let counts = get_vertical_projection(src)?;
    let (rows, cols) = (src.matrix.rows(), src.matrix.cols());
    let mut pic = Mat::new_rows_cols_with_default(rows, cols, opencv::core::CV_8UC1, Scalar::all(255.0))?;
    for c in 0..cols {
        let n = (counts[c as usize] as i32).min(rows);
        for r in (rows - n)..rows {
            *pic.at_2d_mut::<u8>(r, c)? = 0;
        }
    }
    Ok(TransformableMatrix { matrix: pic })
}

/// transfer.rs:459-523 -> omr_rotate.  `flags` is the interpolation flag as the reference passes it
/// (0 = INTER_NEAREST, the numeric value of WARP_POLAR_LINEAR; 1 = INTER_LINEAR); only BORDER_CONSTANT exists.
pub fn rotate_mat(
    src: &TransformableMatrix,
    angle: f64,
    scale: f64,
    flags: i32,
    border_mode: i32,
    border_value: Scalar,
    clip_strategy: RotateClipStrategy,
) -> Result<TransformableMatrix, opencv::Error> {
    if border_mode != opencv::core::BORDER_CONSTANT {
        return Err(opencv::Error::new(ffi::OMR_ERR_NOTIMPL, String::from("only BORDER_CONSTANT is implemented")));
    }
    let border = border_bytes(border_value);
    let mut out = ffi::OmrImageOwned::empty();
    check(unsafe { ffi::omr_rotate(&view(&src.matrix)?, angle, scale, flags, border.as_ptr(), clip_strategy.to_abi(), &mut out) })?;
    Ok(TransformableMatrix { matrix: into_mat(out)? })
}

/// transfer.rs:527-536 -> (std-dev of the vertical projection, std-dev of the horizontal projection).
pub fn get_projection_standard_deviations(src: &TransformableMatrix) -> Result<(f64, f64), opencv::Error> {
    let (mut v_sd, mut h_sd) = (0.0f64, 0.0f64);
    check(unsafe { ffi::omr_get_projection_standard_deviations(&view(&src.matrix)?, &mut v_sd, &mut h_sd) })?;
    Ok((v_sd, h_sd))
}
