// oracle/refkit: stand-in for <opencv2/stitching/detail/blenders.hpp> (see ../../../refkit.h):
// the feather blender over the whole mosaic, forwarding to the oracle's restatement
// (amo::Feather in amo_cvlike.h; NOT pinned by this build).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_OPENCV2_BLENDERS_HPP_
#define ORACLE_REFKIT_OPENCV2_BLENDERS_HPP_

#include <opencv2/highgui/highgui.hpp>

namespace cv {
namespace detail {

class Blender {
 public:
  enum { NO, FEATHER, MULTI_BAND };
  static Ptr<Blender> createDefault(int /*type*/, bool /*try_gpu*/ = false) {
    return Ptr<Blender>(new Blender());
  }
  void prepare(const Rect& dst_roi) { f_.prepare(dst_roi.width, dst_roi.height); }
  // img: CV_16SC3, mask: CV_8U, both the size of the mosaic, top-left corner (0, 0)
  void feed(const Mat& img, const Mat& mask, const Point& /*tl*/) {
    const size_t n = static_cast<size_t>(f_.w) * f_.h;
    std::vector<int16_t> pixels(3 * n);
    std::vector<uint8_t> m(n);
    for (int r = 0; r < f_.h; ++r) {
      std::memcpy(&pixels[3 * static_cast<size_t>(r) * f_.w], img.ptr<int16_t>(r), 6 * static_cast<size_t>(f_.w));
      std::memcpy(&m[static_cast<size_t>(r) * f_.w], mask.ptr<uint8_t>(r), static_cast<size_t>(f_.w));
    }
    f_.feed(pixels, m);
  }
  void blend(Mat& dst, Mat& dst_mask) {
    std::vector<int16_t> pixels;
    std::vector<uint8_t> m;
    f_.blend(&pixels, &m);
    Mat out(f_.h, f_.w, CV_16SC3), out_mask(f_.h, f_.w, CV_8U);
    std::memcpy(out.data, pixels.data(), pixels.size() * sizeof(int16_t));
    std::memcpy(out_mask.data, m.data(), m.size());
    dst = out;
    dst_mask = out_mask;
  }

 private:
  amo::Feather f_;
};

}  // namespace detail
}  // namespace cv

#endif  // ORACLE_REFKIT_OPENCV2_BLENDERS_HPP_
