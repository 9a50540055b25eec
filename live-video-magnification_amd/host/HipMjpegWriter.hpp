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
#include <cstring>
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

// The other direction (source/FileSource.cpp:99 `cap_.read(frame)` on an AVI / Motion-JPEG file): the frames as they lie in the file, for
// lvm_mjpeg_decode_device / lvm_export_mjpeg_frames -- the decoder runs on the GPU, this class only finds the chunks.  AVI 1.0 files
// ('00dc' / '00db' chunks in LIST 'movi', with or without 'idx1'; what this file's writer, OpenCV and FFmpeg write below 4 GiB).
class MjpegAviReader {
public:
    MjpegAviReader() = default;
    ~MjpegAviReader() { close(); }
    MjpegAviReader(const MjpegAviReader&) = delete;
    MjpegAviReader& operator=(const MjpegAviReader&) = delete;

    bool open(const std::string& path) {
        close();
        f_ = std::fopen(path.c_str(), "rb");
        if (!f_) return false;
        std::uint8_t hd[12];
        if (!at(0, hd, 12) || std::memcmp(hd, "RIFF", 4) != 0 || std::memcmp(hd + 8, "AVI ", 4) != 0) { close(); return false; }
        const std::uint64_t end = 8ull + le32(hd + 4);
        bool is_mjpg = false;
        for (std::uint64_t p = 12; p + 8 <= end;) {
            std::uint8_t ch[12];
            if (!at(p, ch, 8)) break;
            const std::uint64_t n = le32(ch + 4);
            if (std::memcmp(ch, "LIST", 4) == 0 && at(p + 8, ch + 8, 4)) {
                if (std::memcmp(ch + 8, "hdrl", 4) == 0) is_mjpg = parse_hdrl(p + 12, p + 8 + n);
                else if (std::memcmp(ch + 8, "movi", 4) == 0) scan_movi(p + 12, p + 8 + n);
            }
            p += 8 + n + (n & 1);
        }
        if (!is_mjpg || frames_.empty() || w_ <= 0 || h_ <= 0) { close(); return false; }
        return true;
    }
    void close() { if (f_) std::fclose(f_); f_ = nullptr; frames_.clear(); w_ = h_ = 0; fps_ = 0.0; strl_seen_ = 0; vid_stream_ = -1; }
    bool isOpened() const { return f_ != nullptr; }
    int width() const { return w_; }
    int height() const { return h_; }
    double fps() const { return fps_; }
    std::size_t frames() const { return frames_.size(); }
    std::size_t frame_bytes(std::size_t k) const { return k < frames_.size() ? frames_[k].bytes : 0; }
    // frame k (a complete JPEG) into dst (frame_bytes(k) bytes)
    bool read(std::size_t k, std::uint8_t* dst) { return k < frames_.size() && at(frames_[k].offset, dst, frames_[k].bytes); }

private:
    struct Chunk { std::uint64_t offset; std::uint32_t bytes; };
    static std::uint32_t le32(const std::uint8_t* p) { return (std::uint32_t)p[0] | ((std::uint32_t)p[1] << 8) | ((std::uint32_t)p[2] << 16) | ((std::uint32_t)p[3] << 24); }
    bool at(std::uint64_t pos, void* dst, std::size_t n) { return f_ && std::fseek(f_, (long)pos, SEEK_SET) == 0 && std::fread(dst, 1, n, f_) == n; }
    bool parse_hdrl(std::uint64_t p, std::uint64_t end, int depth = 0) {
        bool mjpg = false;
        while (p + 8 <= end) {
            std::uint8_t ch[12];
            if (!at(p, ch, 8)) break;
            const std::uint64_t n = le32(ch + 4);
            if (std::memcmp(ch, "avih", 4) == 0 && n >= 40) {
                std::uint8_t a[40];
                if (at(p + 8, a, 40)) { const std::uint32_t usec = le32(a); if (usec) fps_ = 1e6 / usec; w_ = (int)le32(a + 32); h_ = (int)le32(a + 36); }
            } else if (std::memcmp(ch, "LIST", 4) == 0 && at(p + 8, ch + 8, 4) && std::memcmp(ch + 8, "strl", 4) == 0) {
                ++strl_seen_;                                                                    // stream number = position of its 'strl' list
                if (depth < 4) mjpg = parse_hdrl(p + 12, p + 8 + n < end ? p + 8 + n : end, depth + 1) || mjpg;
            } else if (std::memcmp(ch, "strh", 4) == 0 && n >= 32) {
                std::uint8_t a[32];
                if (at(p + 8, a, 32) && std::memcmp(a, "vids", 4) == 0) {
                    const std::uint32_t scale = le32(a + 20), rate = le32(a + 24);
                    if (scale && rate) fps_ = (double)rate / (double)scale;
                    if (std::memcmp(a + 4, "MJPG", 4) == 0 || std::memcmp(a + 4, "mjpg", 4) == 0) mjpg = true;
                    if (vid_stream_ < 0) vid_stream_ = strl_seen_ > 0 ? strl_seen_ - 1 : 0;     // the FIRST video stream is the one read
                }
            } else if (std::memcmp(ch, "strf", 4) == 0 && n >= 20) {
                std::uint8_t a[20];
                if (at(p + 8, a, 20) && (std::memcmp(a + 16, "MJPG", 4) == 0 || std::memcmp(a + 16, "mjpg", 4) == 0)) { mjpg = true; w_ = (int)le32(a + 4); h_ = (int)le32(a + 8); }
            }
            p += 8 + n + (n & 1);
        }
        return mjpg;
    }
    // chunks '##dc' / '##db' of the video stream only (## = its two-digit stream number: another stream's chunks -- a second video
    // track, audio mislabelled by a broken muxer -- are not frames of this one); 'rec ' groups nest at most a few levels in any real file
    void scan_movi(std::uint64_t p, std::uint64_t end, int depth = 0) {
        const int vs = vid_stream_ < 0 ? 0 : vid_stream_;
        const std::uint8_t d0 = (std::uint8_t)('0' + (vs / 10) % 10), d1 = (std::uint8_t)('0' + vs % 10);
        while (p + 8 <= end) {
            std::uint8_t ch[12];
            if (!at(p, ch, 8)) break;
            const std::uint64_t n = le32(ch + 4);
            if (std::memcmp(ch, "LIST", 4) == 0) { if (depth < 4) scan_movi(p + 12, p + 8 + n < end ? p + 8 + n : end, depth + 1); }
            else if (ch[0] == d0 && ch[1] == d1 && ch[2] == 'd' && (ch[3] == 'c' || ch[3] == 'b') && n >= 4 && p + 8 + n <= end) frames_.push_back({p + 8, (std::uint32_t)n});
            p += 8 + n + (n & 1);
        }
    }
    std::FILE* f_ = nullptr;
    int w_ = 0, h_ = 0;
    double fps_ = 0.0;
    std::vector<Chunk> frames_;
    int strl_seen_ = 0, vid_stream_ = -1;
};

}  // namespace lvm
