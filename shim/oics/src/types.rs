//! Plain option enums of the public API (reference: packages/lib/src/types.rs:1-11).

/// Container format for `TransformableMatrix::im_write`.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ImageFormat {
    JPEG,
    PNG,
    WEBP,
}

/// Canvas policy of `transfer::rotate_mat`: keep the source size, or grow the canvas so the whole
/// rotated sheet stays visible.  Maps to OMR_CLIP_DEFAULT / OMR_CLIP_CONTAIN.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum RotateClipStrategy {
    DEFAULT,
    CONTAIN,
}

impl RotateClipStrategy {
    pub(crate) fn to_abi(self) -> i32 {
        match self {
            RotateClipStrategy::DEFAULT => crate::ffi::OMR_CLIP_DEFAULT,
            RotateClipStrategy::CONTAIN => crate::ffi::OMR_CLIP_CONTAIN,
        }
    }
}
