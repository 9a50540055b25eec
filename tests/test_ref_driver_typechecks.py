"""oracle/ref_driver.cpp -- the C entry points over the REAL reference stage that `make -C oracle ref_full` builds where OpenCV 4 exists
(tools/pin_with_opencv.sh) -- type-checked HERE, where OpenCV does not exist: `g++ -std=c++20 -fsyntax-only` over the driver and every
reference header it pulls in (MagnificationProcessor.hpp -> MagnifyCore.hpp with its inline magnifyMotion / magnifyColor /
magnifyRiesz, ComplexMat.hpp, RieszPyramid.hpp, SpatialFilter.hpp, TemporalFilter.hpp, core/Frame.hpp) against a DECLARATION-ONLY
stand-in for <opencv2/core.hpp> / <opencv2/imgproc.hpp>.  What it proves: the driver's use of the reference's types (Frame fields,
ProcessorConfig / MagnificationParams members, MagnificationProcessor::process / reset, the enum casts) and of cv::Mat compiles;
what it cannot prove: linking and OpenCV's arithmetic -- that is tools/pin_with_opencv.sh's job.  Skipped where the reference
checkout is absent (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"

CV_DECLS = r'''
#pragma once
// DECLARATION-ONLY stand-in for <opencv2/core.hpp> / <opencv2/imgproc.hpp> (TEST INFRASTRUCTURE, tests/test_ref_driver_typechecks.py):
// just enough of the cv:: surface for `g++ -fsyntax-only` to type-check oracle/ref_driver.cpp together with the reference headers it
// includes (MagnificationProcessor.hpp -> MagnifyCore.hpp and its inline magnifyMotion / magnifyColor / magnifyRiesz, ComplexMat.hpp,
// RieszPyramid.hpp, SpatialFilter.hpp, TemporalFilter.hpp, core/Frame.hpp).  Nothing is defined: it cannot link, and is not meant to.
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>
#define CV_PI 3.1415926535897932384626433832795
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5
#define CV_32FC3 21
namespace cv {
struct Size { int width = 0, height = 0; Size(); Size(int w, int h); bool operator==(const Size&) const; bool operator!=(const Size&) const; };
struct Vec3f { float v[3]; Vec3f(); Vec3f(float a, float b, float c); float& operator[](int i); const float& operator[](int i) const; };
struct MatStep { size_t v; operator size_t() const; };
class Mat;
class MatExpr { public: operator Mat() const; };
class Mat {
public:
    Mat(); Mat(int rows, int cols, int type); Mat(int rows, int cols, int type, void* data, size_t step = 0); Mat(Size s, int type);
    Mat(const MatExpr&); Mat& operator=(const MatExpr&);
    void convertTo(Mat& dst, int rtype, double alpha = 1, double beta = 0) const;
    void copyTo(Mat& dst) const;
    Mat clone() const;
    Size size() const;
    int channels() const; int type() const; bool empty() const;
    unsigned char* ptr(int y = 0); const unsigned char* ptr(int y = 0) const;
    template <class T> T& at(int r, int c); template <class T> const T& at(int r, int c) const;
    unsigned char* data; int rows, cols; MatStep step;
};
MatExpr operator*(const Mat&, double); MatExpr operator*(double, const Mat&); MatExpr operator*(const MatExpr&, double);
MatExpr operator+(const Mat&, const Mat&); MatExpr operator-(const Mat&, const Mat&);
MatExpr operator+(const MatExpr&, const Mat&); MatExpr operator+(const Mat&, const MatExpr&); MatExpr operator+(const MatExpr&, const MatExpr&);
MatExpr operator-(const MatExpr&, const Mat&); MatExpr operator-(const Mat&, const MatExpr&); MatExpr operator-(const MatExpr&, const MatExpr&);
enum ColorConversionCodes { COLOR_BGR2Lab = 44, COLOR_Lab2BGR = 56, COLOR_BGR2GRAY = 6, COLOR_GRAY2BGR = 8 };
void cvtColor(const Mat& src, Mat& dst, int code);
void split(const Mat& m, Mat* planes); void split(const Mat& m, std::vector<Mat>& planes);
void merge(const Mat* planes, size_t n, Mat& dst); void merge(const std::vector<Mat>& planes, Mat& dst);
void add(const Mat& a, const Mat& b, Mat& dst); void subtract(const Mat& a, const Mat& b, Mat& dst);
void multiply(const Mat& a, const Mat& b, Mat& dst, double scale = 1); void multiply(const Mat& a, double b, Mat& dst);
void divide(const Mat& a, const Mat& b, Mat& dst, double scale = 1); void divide(const Mat& a, double b, Mat& dst);
void minMaxLoc(const Mat& src, double* minVal, double* maxVal = nullptr);
}  // namespace cv
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout absent")
def test_ref_driver_typechecks_against_the_reference_headers(tmp_path):
    inc = tmp_path / "opencv2"
    inc.mkdir()
    (inc / "core.hpp").write_text(CV_DECLS)
    (inc / "imgproc.hpp").write_text("#pragma once\n#include <opencv2/core.hpp>\n")
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-I", str(tmp_path), "-I", REF, os.path.join(ROOT, "oracle", "ref_driver.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # and the check is not vacuous: a driver that misuses the reference's types must fail
    bad = tmp_path / "bad.cpp"
    bad.write_text(open(os.path.join(ROOT, "oracle", "ref_driver.cpp")).read().replace("cfg.magnification.levels = prm->levels;", "cfg.magnification.no_such_field = prm->levels;"))
    r = subprocess.run(cmd[:-1] + [str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "no_such_field" in r.stderr
