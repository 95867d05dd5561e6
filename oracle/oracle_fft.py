"""CPU restatement of the reference's FFT deskew path -- numpy.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  "parity unpinned", and looser than the other rows:
the arithmetic lives in OpenCV 4.6.0's dft / magnitude / log / convertTo (not under /root/reference),
whose float32 rounding (mixed-radix butterflies, table-driven log, SIMD multiply-add) cannot be restated
bit for bit from the published algorithm.  This file states the *definition* -- a double-precision DFT
rounded once to float32, then the reference's float32 operation sequence -- so a correct float32
implementation agrees with it to about 1 grey level of the 8-bit spectrum pictures; everything after the
picture (Canny, HoughLinesP, votes) is the exact restatement of oracle_hough.c.

  fft.rs:42-66   fft_complex: merge(image, 0) -> dft(COMPLEX_INPUT | COMPLEX_OUTPUT | SCALE) -> split -> fft_shift
  fft.rs:68-88   fft_shift: quadrant swap with cx = cols / 2, cy = rows / 2 (an odd last row / column stays put)
  fft.rs:90-106  correction: (x - min) * (1 / (max - min))
  fft.rs:108-122 fft_magnitude = correction(|F|) * 255;  fft_magnitude_log = correction(log(that + 1/255))
  fft.rs:124-141 get_fft_image -> (magnitude, magnitude_log) as 8-bit via convert_to(CV_8UC1, 255)
  fft.rs:145-256 get_angle_with_fft;  omr.rs:304-337 get_result_from_fourier_transform
"""
import numpy as np

from . import oracle as orc

f32 = np.float32


def fft_shift(a):
    """fft.rs:68-88"""
    a = a.copy()
    rows, cols = a.shape
    cx, cy = cols // 2, rows // 2
    q0 = a[0:cy, 0:cx].copy()
    q1 = a[0:cy, cx:2 * cx].copy()
    q2 = a[cy:2 * cy, 0:cx].copy()
    q3 = a[cy:2 * cy, cx:2 * cx].copy()
    a[0:cy, 0:cx] = q3
    a[cy:2 * cy, cx:2 * cx] = q0
    a[0:cy, cx:2 * cx] = q2
    a[cy:2 * cy, 0:cx] = q1
    return a


def correction(img):
    """fft.rs:90-106: two convert_to(CV_32F) steps, float32 arithmetic with the double alpha / beta cast to float"""
    mn, mx = float(img.min()), float(img.max())
    c1 = img * f32(1.0) + f32(-mn)
    return (c1 * f32(1.0 / (mx - mn)) + f32(0.0)).astype(f32)


def spectrum(gray):
    """fft.rs:42-66 -> shifted (re, im) float32"""
    g = np.asarray(gray, np.uint8)
    img = g.astype(f32) * f32(1.0 / 255.0) + f32(0.0)
    F = np.fft.fft2(img.astype(np.float64)) / float(g.shape[0] * g.shape[1])
    return fft_shift(F.real.astype(f32)), fft_shift(F.imag.astype(f32))


def to_u8(img, alpha=255.0):
    """convert_to(CV_8UC1, alpha): saturate_cast<uchar>(cvRound(x * alpha)) in float32"""
    v = np.rint(img.astype(f32) * f32(alpha))
    return np.clip(v, 0, 255).astype(np.uint8)


def fft_magnitude(re, im):
    mag = np.sqrt(re * re + im * im).astype(f32)
    return (correction(mag) * f32(255.0) + f32(0.0)).astype(f32)


def get_fft_image(gray):
    """fft.rs:124-141 -> (magnitude_image, magnitude_log_image), both uint8"""
    re, im = spectrum(gray)
    m = fft_magnitude(re, im)
    lg = np.log((m * f32(1.0) + f32(1.0 / 255.0)).astype(f32)).astype(f32)
    return to_u8(m), to_u8(correction(lg))


def vote_fft_rs(lines):
    """fft.rs:197-247: f64 atan2; the inner loop reads line i again (quirk B10), so a line collects
    n - 1 votes when its raw angle is within 0.1 of its folded angle, i.e. lies in [-45, 45], else 0;
    strict '>' from max_votes = 0 keeps the first such line; none (or a single line) -> 0.0."""
    lines = np.asarray(lines, np.float64).reshape(-1, 4)
    n = len(lines)
    average, max_votes = 0.0, 0
    for i in range(n):
        x1, y1, x2, y2 = lines[i]
        raw = (np.arctan2(y2 - y1, x2 - x1) * 180.0) / np.pi
        angle = raw + 90.0 if raw < -45.0 else (raw - 90.0 if raw > 45.0 else raw)
        votes = (n - 1) if abs(raw - angle) < 0.1 else 0
        if votes > max_votes:
            max_votes, average = votes, float(angle)
    return average


def get_angle_with_fft(gray, canny_threshold_1, canny_threshold_2, min_line_length, max_line_gap):
    """fft.rs:145-256 without the debug picture"""
    _, lg = get_fft_image(gray)
    edges = orc.canny(lg, canny_threshold_1, canny_threshold_2)
    lines = orc.hough_lines_p(edges, min_line_length, max_line_gap, threshold=100)
    return vote_fft_rs(lines), lg, edges, lines


def get_result_from_fourier_transform(src_bgr, canny_weak, canny_strong, min_line_length, max_line_gap):
    """omr.rs:304-337: RGB2GRAY -> log spectrum -> Canny(weak, strong) -> get_result_from_edges_detection
    (which runs Canny(50, 150) once more on the edge picture, omr.rs:236-240)"""
    gray = orc.rgb2gray(src_bgr)
    _, lg = get_fft_image(gray)
    edges = orc.canny(lg, canny_weak, canny_strong)
    return orc.get_result_from_edges_detection(edges, min_line_length, max_line_gap), lg, edges
