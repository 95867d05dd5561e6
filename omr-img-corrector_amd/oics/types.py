"""packages/lib/src/types.rs:1-11 and omr.rs:41-50."""
import enum


class ImageFormat(enum.Enum):
    JPEG = 0
    PNG = 1
    WEBP = 2


class RotateClipStrategy(enum.IntEnum):
    DEFAULT = 0
    CONTAIN = 1


class ResultStatus(enum.IntEnum):
    Believed = 0
    NeedCheck = 1
    NotAResult = 2
