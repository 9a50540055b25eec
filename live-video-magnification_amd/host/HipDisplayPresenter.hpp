// HipDisplayPresenter.hpp -- the display half of SURVEY.md 8(f) rank 2: the live path WITHOUT the download.
//
// The reference's live path per frame (one processing thread, one GL thread):
//     ProcessingChain::run                 cur = runChainOnce(chain_, in, *cfg, original)        (processing/ProcessingChain.cpp:43)
//                                          out_->publish({processed = cur, original})            (:46-49)
//     DisplayWidget::paintGL               uploadFrame(*df->processed, texProc_), uploadFrame(*df->original, texOrig_)
//     DisplayWidget::uploadFrame           glTexImage2D / glTexSubImage2D(..., src.data)         (ui/DisplayWidget.cpp:133-152)
// i.e. every processed frame comes back over PCIe into a cv::Mat only to be sent to the GPU again by the GL driver.  With the
// magnifier on the GPU the two frames are already there: this presenter owns two GL PIXEL-UNPACK BUFFERS (one per texture),
// registers them with HIP once (hipGraphicsGLRegisterBuffer), maps them for the duration of one lvm_chain_present call -- whose last
// kernel writes the processed frame, and whose preprocess kernel writes the `original` tap, straight into the mapped pointers -- and
// then lets glTexSubImage2D copy buffer -> texture on the device.  Only the ROI rows of the camera frame cross PCIe.
//
// The core is a template over a traits type (the GL / interop calls), so that it can be compiled and run here -- no GL context
// exists in this image -- against a mock whose "pixel buffer" is a hipMalloc'd buffer (tests/test_host_display.py); GlInteropTraits
// below binds it to OpenGL + hip_gl_interop.h and is compiled (not run) by the same test where <GL/gl.h> exists.
//
// What a traits type provides:
//   struct Buffer;                                        one pixel-unpack buffer + its HIP registration
//   static void  create(Buffer&, std::size_t bytes);      (re)allocate for `bytes`, register with HIP
//   static void  destroy(Buffer&);
//   static std::uint8_t* map(Buffer&);                    device pointer valid until unmap()
//   static void  unmap(Buffer&);
//   static void  upload(Buffer&, Texture&, int w, int h, int channels);    buffer -> texture (DisplayWidget.cpp:133-152 with a bound PBO)
//   struct Texture;
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "lvm.hpp"

namespace lvm {

template <class T>
class DisplayPresenter {
public:
    explicit DisplayPresenter(int device = 0) : mag_(device, 1) {}
    ~DisplayPresenter() { release(); }
    DisplayPresenter(const DisplayPresenter&) = delete;
    DisplayPresenter& operator=(const DisplayPresenter&) = delete;

    struct Shown { int w = 0, h = 0, proc_channels = 0, orig_channels = 0; bool produced = false; };

    // One frame of the live path: runChainOnce on the GPU, both panes into their textures.  Must run on the thread that owns the GL
    // context (DisplayWidget::paintGL's).  Throws lvm::Error on a library failure; the caller's recovery is the reference's
    // (ProcessingChain.cpp:50-58: count it, reset(), show the input).
    Shown present(const std::uint8_t* frame, int w, int h, int channels, std::ptrdiff_t stride, const lvm_preprocess_params& pre,
                  const MagnificationParams& mag, typename T::Texture& tex_proc, typename T::Texture& tex_orig) {
        int ow = 0, oh = 0, och = 0;
        Magnifier::chain_geometry(pre, w, h, channels, &ow, &oh, &och);
        const std::size_t pb = (std::size_t)ow * oh * och, ob = (std::size_t)ow * oh * channels;
        if (pb != proc_bytes_ || ob != orig_bytes_) {            // geometry changed: new buffers (the textures follow in upload())
            release();
            T::create(proc_, pb); T::create(orig_, ob);
            proc_bytes_ = pb; orig_bytes_ = ob; have_ = true;
        }
        int produced = 0, rc = LVM_OK;
        {
            // a buffer left mapped makes the following glTexSubImage2D and the next present() fail: whatever throws between the two
            // maps and the two unmaps (a failing hipGraphicsMapResources on the second buffer, an unmap that throws), both buffers are
            // unmapped when this block is left
            Mapped mp(proc_);
            Mapped mo(orig_);
            const lvm_params c = to_c(mag, 0);
            rc = lvm_chain_present(mag_.handle(), &pre, &c, frame, w, h, channels, stride, mp.ptr, (std::ptrdiff_t)ow * och, mo.ptr,
                                   (std::ptrdiff_t)ow * channels, &produced);
            mo.release(); mp.release();      // (the normal path reports an unmap failure; the destructors only clean up behind an exception)
        }
        if (rc != LVM_OK) throw Error(rc, std::string("lvm: ") + lvm_last_error(mag_.handle()));
        T::upload(proc_, tex_proc, ow, oh, och);
        T::upload(orig_, tex_orig, ow, oh, channels);
        Shown s; s.w = ow; s.h = oh; s.proc_channels = och; s.orig_channels = channels; s.produced = produced != 0;
        return s;
    }
    void reset() { mag_.reset(); }

private:
    struct Mapped {                      // one mapped buffer; unmapped on scope exit, never throwing from the destructor
        typename T::Buffer& b; std::uint8_t* ptr = nullptr; bool held = false;
        explicit Mapped(typename T::Buffer& buf) : b(buf) { ptr = T::map(b); held = true; }
        void release() { if (held) { held = false; T::unmap(b); } }
        ~Mapped() { if (held) { try { T::unmap(b); } catch (...) {} } }
        Mapped(const Mapped&) = delete; Mapped& operator=(const Mapped&) = delete;
    };
    void release() {
        if (have_) { T::destroy(proc_); T::destroy(orig_); have_ = false; }
        proc_bytes_ = orig_bytes_ = 0;
    }
    Magnifier mag_;
    typename T::Buffer proc_{}, orig_{};
    std::size_t proc_bytes_ = 0, orig_bytes_ = 0;
    bool have_ = false;
};

}  // namespace lvm

#ifdef LVM_WITH_GL_INTEROP
// OpenGL + HIP interop binding (needs a current GL context on the calling thread; Qt's QOpenGLFunctions resolve the same entry
// points in the reference, ui/DisplayWidget.cpp).  Compiled where <GL/gl.h> / <GL/glext.h> and <hip/hip_gl_interop.h> exist.
#define GL_GLEXT_PROTOTYPES 1
#include <GL/gl.h>
#include <GL/glext.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <hip/hip_gl_interop.h>

namespace lvm {

struct GlInteropTraits {
    struct Buffer { GLuint pbo = 0; hipGraphicsResource_t res = nullptr; std::size_t bytes = 0; };
    struct Texture { GLuint id = 0; int w = 0, h = 0, channels = 0; };       // = DisplayWidget::Tex (ui/DisplayWidget.hpp)
    static void check(hipError_t e, const char* what) { if (e != hipSuccess) throw Error(LVM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
    static void create(Buffer& b, std::size_t bytes) {
        glGenBuffers(1, &b.pbo);
        glBindBuffer(GL_PIXEL_UNPACK_BUFFER, b.pbo);
        glBufferData(GL_PIXEL_UNPACK_BUFFER, (GLsizeiptr)bytes, nullptr, GL_STREAM_DRAW);
        glBindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
        check(hipGraphicsGLRegisterBuffer(&b.res, b.pbo, hipGraphicsRegisterFlagsWriteDiscard), "hipGraphicsGLRegisterBuffer");
        b.bytes = bytes;
    }
    static void destroy(Buffer& b) {
        if (b.res) (void)hipGraphicsUnregisterResource(b.res);
        if (b.pbo) glDeleteBuffers(1, &b.pbo);
        b = Buffer{};
    }
    static std::uint8_t* map(Buffer& b) {
        check(hipGraphicsMapResources(1, &b.res, nullptr), "hipGraphicsMapResources");
        void* p = nullptr; std::size_t n = 0;
        check(hipGraphicsResourceGetMappedPointer(&p, &n, b.res), "hipGraphicsResourceGetMappedPointer");
        return static_cast<std::uint8_t*>(p);
    }
    static void unmap(Buffer& b) { check(hipGraphicsUnmapResources(1, &b.res, nullptr), "hipGraphicsUnmapResources"); }
    // DisplayWidget::uploadFrame (ui/DisplayWidget.cpp:133-152) with the pixel-unpack buffer bound: the "pointer" argument of
    // glTex(Sub)Image2D is an offset into the buffer, the copy never leaves the device
    static void upload(Buffer& b, Texture& t, int w, int h, int channels) {
        const GLint internal = channels == 1 ? GL_R8 : GL_RGB8;
        const GLenum fmt = channels == 1 ? GL_RED : GL_RGB;
        glBindTexture(GL_TEXTURE_2D, t.id);
        glBindBuffer(GL_PIXEL_UNPACK_BUFFER, b.pbo);
        glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
        glPixelStorei(GL_UNPACK_ROW_LENGTH, w);
        if (w != t.w || h != t.h || channels != t.channels) {
            glTexImage2D(GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, GL_UNSIGNED_BYTE, nullptr);
            t.w = w; t.h = h; t.channels = channels;
        } else {
            glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, w, h, fmt, GL_UNSIGNED_BYTE, nullptr);
        }
        glPixelStorei(GL_UNPACK_ROW_LENGTH, 0);
        glBindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
        glBindTexture(GL_TEXTURE_2D, 0);
    }
};
using GlDisplayPresenter = DisplayPresenter<GlInteropTraits>;

}  // namespace lvm
#endif  // LVM_WITH_GL_INTEROP
