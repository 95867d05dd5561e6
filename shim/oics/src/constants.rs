//! Named imread flags (reference: packages/lib/src/constants.rs:3-39).  Host-side codec plumbing only.
use opencv::imgcodecs as ic;

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ImReadFlags {
    AnyColor,
    AnyDepth,
    Color,
    Grayscale,
    IgnoreOrientation,
    LoadGDal,
    ReducedColor2,
    ReducedColor4,
    ReducedColor8,
    ReducedGrayscale2,
    ReducedGrayscale4,
    ReducedGrayscale8,
    Unchanged,
}

impl ImReadFlags {
    /// The OpenCV integer behind a flag (`ImReadFlags::from(ImReadFlags::Color)`).
    pub fn from(flag: Self) -> i32 {
        const TABLE: [(ImReadFlags, i32); 13] = [
            (ImReadFlags::AnyColor, ic::IMREAD_ANYCOLOR),
            (ImReadFlags::AnyDepth, ic::IMREAD_ANYDEPTH),
            (ImReadFlags::Color, ic::IMREAD_COLOR),
            (ImReadFlags::Grayscale, ic::IMREAD_GRAYSCALE),
            (ImReadFlags::IgnoreOrientation, ic::IMREAD_IGNORE_ORIENTATION),
            (ImReadFlags::LoadGDal, ic::IMREAD_LOAD_GDAL),
            (ImReadFlags::ReducedColor2, ic::IMREAD_REDUCED_COLOR_2),
            (ImReadFlags::ReducedColor4, ic::IMREAD_REDUCED_COLOR_4),
            (ImReadFlags::ReducedColor8, ic::IMREAD_REDUCED_COLOR_8),
            (ImReadFlags::ReducedGrayscale2, ic::IMREAD_REDUCED_GRAYSCALE_2),
            (ImReadFlags::ReducedGrayscale4, ic::IMREAD_REDUCED_GRAYSCALE_4),
            (ImReadFlags::ReducedGrayscale8, ic::IMREAD_REDUCED_GRAYSCALE_8),
            (ImReadFlags::Unchanged, ic::IMREAD_UNCHANGED),
        ];
        TABLE.iter().find(|(f, _)| *f == flag).map(|(_, v)| *v).unwrap()
    }
}
