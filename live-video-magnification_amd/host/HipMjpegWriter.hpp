// HipMjpegWriter.hpp -- the container half of the device-side Motion-JPEG export (encode).
//
// The reference writes its export through cv::VideoWriter (src/export/Exporter.cpp:92-118 openWriter, :259 writer.write(canvas)); with
// ExportFormat::AviMjpg -- and as the fallback of the other two formats (:117) -- that is an AVI file of JPEG frames, encoded on the host
// one canvas at a time.  lvm_export_frames_mjpeg hands back the frames ALREADY encoded (include/lvm_hip.h): what is left of the writer is
// the RIFF container, i.e. this class.  Same life cycle as the cv::VideoWriter it replaces:
//     open(path, w, h, fps)   <->  writer.open(path, fourcc('M','J','P','G'), fps, size, true)
//     write(jpeg, bytes)      <->  writer.write(canvas)              (one '00dc' chunk per frame, every frame a key frame)
//     close()                 <->  writer.release()                  (index 'idx1', sizes and frame count patched into the headers)
// AVI 1.0 (one RIFF chunk, < 4 GiB: write() returns false when the next frame would not fit -- the caller finalises and starts the next
// file, as it would on any other write failure, Exporter.cpp:261-289).  No OpenCV, no HIP: plain C++17 + <cstdio>.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace lvm {

class MjpegAviWriter {
public:
    MjpegAviWriter() = default;
    ~MjpegAviWriter() { close(); }
    MjpegAviWriter(const MjpegAviWriter&) = delete;
    MjpegAviWriter& operator=(const MjpegAviWriter&) = delete;

    bool open(const std::string& path, int width, int height, double fps) {
        close();
        if (width <= 0 || height <= 0 || !(fps > 0.0)) return false;
        f_ = std::fopen(path.c_str(), "wb");
        if (!f_) return false;
        w_ = width; h_ = height; frames_ = 0; max_chunk_ = 0; index_.clear(); ok_ = true; pos_ = 0;
        // frame rate as a fraction: integers stay integers (30 -> 30 / 1), everything else in thousandths (29.97 -> 29970 / 1000)
        const double r = fps * 1000.0;
        if (fps == (double)(std::uint32_t)fps) { rate_ = (std::uint32_t)fps; scale_ = 1; }
        else { rate_ = (std::uint32_t)(r + 0.5); scale_ = 1000; }
        header(0, 0, 0);                                   // placeholders, rewritten by close()
        return ok_;
    }
    bool isOpened() const { return f_ != nullptr; }
    // one encoded frame (a complete JPEG: SOI .. EOI)
    bool write(const std::uint8_t* jpeg, std::size_t bytes) {
        if (!f_ || !ok_ || !jpeg || bytes < 4 || jpeg[0] != 0xFF || jpeg[1] != 0xD8) return false;
        const std::uint64_t after = pos_ + 8 + bytes + (bytes & 1) + 16ull * (index_.size() + 1) + 8;
        if (after > 0xFFFFFFF0ull || bytes > 0x7FFFFFFFull) return false;                     // AVI 1.0: one RIFF chunk
        index_.push_back({(std::uint32_t)(pos_ - movi_start_), (std::uint32_t)bytes});         // offset of the chunk header from 'movi'
        fourcc("00dc"); u32((std::uint32_t)bytes);
        put(jpeg, bytes);
        if (bytes & 1) { const std::uint8_t z = 0; put(&z, 1); }                               // chunks are word-aligned
        if (bytes > max_chunk_) max_chunk_ = (std::uint32_t)bytes;
        ++frames_;
        return ok_;
    }
    std::uint32_t frames() const { return frames_; }
    bool close() {
        if (!f_) return true;
        const std::uint64_t movi_end = pos_, movi_start = movi_start_;
        fourcc("idx1"); u32((std::uint32_t)(16 * index_.size()));
        for (const Entry& e : index_) { fourcc("00dc"); u32(0x10); u32(e.offset); u32(e.bytes); }   // AVIIF_KEYFRAME
        const std::uint64_t total = pos_;
        if (ok_ && std::fseek(f_, 0, SEEK_SET) == 0) {
            pos_ = 0;
            header((std::uint32_t)(total - 8), (std::uint32_t)(movi_end - movi_start), frames_);
        } else ok_ = false;
        const bool flushed = std::fclose(f_) == 0;
        f_ = nullptr;
        return ok_ && flushed;
    }

private:
    struct Entry { std::uint32_t offset, bytes; };
    void put(const void* p, std::size_t n) { if (ok_ && std::fwrite(p, 1, n, f_) != n) ok_ = false; pos_ += n; }
    void u32(std::uint32_t v) { const std::uint8_t b[4] = {(std::uint8_t)v, (std::uint8_t)(v >> 8), (std::uint8_t)(v >> 16), (std::uint8_t)(v >> 24)}; put(b, 4); }
    void u16(std::uint16_t v) { const std::uint8_t b[2] = {(std::uint8_t)v, (std::uint8_t)(v >> 8)}; put(b, 2); }
    void fourcc(const char* s) { put(s, 4); }
    // RIFF 'AVI ' { LIST 'hdrl' { avih, LIST 'strl' { strh, strf } }, LIST 'movi' ...
    void header(std::uint32_t riff_size, std::uint32_t movi_size, std::uint32_t nframes) {
        const std::uint32_t usec = (std::uint32_t)(1e6 * (double)scale_ / (double)rate_ + 0.5);
        fourcc("RIFF"); u32(riff_size); fourcc("AVI ");
        fourcc("LIST"); u32(4 + (8 + 56) + (8 + 4 + (8 + 56) + (8 + 40))); fourcc("hdrl");
        fourcc("avih"); u32(56);
        u32(usec); u32((std::uint32_t)((double)max_chunk_ * (double)rate_ / (double)scale_)); u32(0); u32(0x10);      // AVIF_HASINDEX
        u32(nframes); u32(0); u32(1); u32(max_chunk_); u32((std::uint32_t)w_); u32((std::uint32_t)h_);
        u32(0); u32(0); u32(0); u32(0);
        fourcc("LIST"); u32(4 + (8 + 56) + (8 + 40)); fourcc("strl");
        fourcc("strh"); u32(56);
        fourcc("vids"); fourcc("MJPG"); u32(0); u16(0); u16(0); u32(0); u32(scale_); u32(rate_); u32(0); u32(nframes); u32(max_chunk_); u32(0xFFFFFFFFu); u32(0);
        u16(0); u16(0); u16((std::uint16_t)w_); u16((std::uint16_t)h_);
        fourcc("strf"); u32(40);
        u32(40); u32((std::uint32_t)w_); u32((std::uint32_t)h_); u16(1); u16(24); fourcc("MJPG"); u32((std::uint32_t)w_ * (std::uint32_t)h_ * 3u); u32(0); u32(0); u32(0); u32(0);
        fourcc("LIST"); u32(movi_size); fourcc("movi");
        movi_start_ = pos_ - 4;
    }
    std::FILE* f_ = nullptr;
    bool ok_ = true;
    int w_ = 0, h_ = 0;
    std::uint32_t rate_ = 30, scale_ = 1, frames_ = 0, max_chunk_ = 0;
    std::uint64_t pos_ = 0, movi_start_ = 0;
    std::vector<Entry> index_;
};

}  // namespace lvm
