// bzip2javapi.hpp -- C++ mirror of the reference's public compressor surface (Swift module
// bzip2JavApi) on top of the C ABI in bz2hip.h.  Same class names, method names, argument
// meaning and error behaviour as the Swift sources, so that tests read like the reference's.
// (The Swift toolchain is absent from this image; INTEGRATION.md shows the Swift shim.)
//
//   java::io::OutputStream            stand-in for JavApi's java.io.OutputStream (the sink)
//   BZip2BitOutputStream              S/BZip2BitOutputStream.swift:31-128   (host bit cursor)
//   BZip2BlockCompressor              S/BZip2BlockCompressor.swift:37-321   (GPU at close())
//   BZip2OutputStream                 S/BZip2OutputStream.swift:31-201      (GPU streaming ABI)
//   java::io::InputStream             stand-in for JavApi's java.io.InputStream (the source)
//   BZip2InputStream                  S/BZip2InputStream.swift:45-258       (GPU decoder, whole stream)
//
// Header-only; link with -lbz2hip.  No compute happens on the host: the block compressor keeps
// only the reference's acceptance rule (two counters) to answer write() -> Bool.
#pragma once
#include "bz2hip.h"

#include <algorithm>
#include <cstdint>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

namespace java { namespace io {
// java.io.OutputStream contract used by the reference: write(int) takes the low 8 bits
// (S/BZip2BitOutputStream.swift:58,93), flush(), close().
struct OutputStream {
    virtual ~OutputStream() {}
    virtual void write(int b) = 0;
    virtual void flush() {}
    virtual void close() {}
};
struct ByteArrayOutputStream : OutputStream {
    std::vector<uint8_t> buf;
    bool closed = false;
    void write(int b) override { buf.push_back((uint8_t)(b & 0xff)); }
    void close() override { closed = true; }
};
// java.io.InputStream contract used by the reference: read() returns a byte or -1
// (S/BZip2BitInputStream.swift:60,89,120), close().
struct InputStream {
    virtual ~InputStream() {}
    virtual int read() = 0;
    virtual void close() {}
};
struct ByteArrayInputStream : InputStream {
    std::vector<uint8_t> buf;
    size_t pos = 0;
    bool closed = false;
    explicit ByteArrayInputStream(std::vector<uint8_t> b) : buf(std::move(b)) {}
    int read() override { return pos < buf.size() ? (int)buf[pos++] : -1; }
    void close() override { closed = true; }
};
}} // namespace java::io

namespace bzip2javapi {

// S/bzip2JavApi.swift:25-29: enum BZip2Exception { IOException(msg), BZip2Exception(msg) }
struct BZip2Exception : std::runtime_error {
    enum Kind { IOException, BZip2ExceptionKind } kind;
    BZip2Exception(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};
// java.lang.Throwable.IllegalArgumentException (S/BZip2OutputStream.swift:173-175)
struct IllegalArgumentException : std::invalid_argument {
    explicit IllegalArgumentException(const std::string &m) : std::invalid_argument(m) {}
};

namespace detail {
struct Ctx {
    bz2hip_ctx *h = nullptr;
    Ctx() {
        bz2hip_opts o{-1, 0, 0, 0};
        if (bz2hip_create(&h, &o) != BZ2HIP_OK)
            throw BZip2Exception(BZip2Exception::IOException,
                                 std::string("bz2hip_create: ") + bz2hip_last_error(nullptr));
    }
    ~Ctx() { bz2hip_destroy(h); }
    Ctx(const Ctx &) = delete;
    Ctx &operator=(const Ctx &) = delete;
};
} // namespace detail

// ---- S/BZip2BitOutputStream.swift:31-128 (32-bit accumulator semantics)
class BZip2BitOutputStream {
    java::io::OutputStream &outputStream;
    uint32_t bitBuffer = 0;
    int bitCount = 0;

  public:
    explicit BZip2BitOutputStream(java::io::OutputStream &out) : outputStream(out) {} // :125-127
    void writeBoolean(bool value) {                                                   // :53-65
        int bc = bitCount + 1;
        uint32_t bb = bitBuffer | ((value ? 1u : 0u) << (32 - bc));
        if (bc == 8) {
            outputStream.write((int)(bb >> 24));
            bb = 0;
            bc = 0;
        }
        bitBuffer = bb;
        bitCount = bc;
    }
    void writeUnary(int value) {                                                      // :72-78
        while (value-- > 0) writeBoolean(true);
        writeBoolean(false);
    }
    void writeBits(int count, uint32_t value) {                                       // :86-100
        if (count == 0) return;
        int bc = bitCount;
        uint32_t bb = bitBuffer | ((value << (32 - count)) >> bc);
        bc += count;
        while (bc >= 8) {
            outputStream.write((int)(bb >> 24));
            bb <<= 8;
            bc -= 8;
        }
        bitBuffer = bb;
        bitCount = bc;
    }
    void writeInteger(uint32_t value) {                                               // :107-110
        writeBits(16, (value >> 16) & 0xffff);
        writeBits(16, value & 0xffff);
    }
    void flush() {                                                                    // :116-120
        if (bitCount > 0) writeBits(8 - bitCount, 0);
    }
};

// ---- S/BZip2BlockCompressor.swift:37-321
class BZip2BlockCompressor {
    BZip2BitOutputStream &bitOutputStream;
    detail::Ctx ctx;
    int level;
    std::vector<uint8_t> raw;  // bytes accepted into this block (RLE1/CRC/BWT/... run on the GPU)
    int blockLength = 0;       // flushed RLE1 length, tracked only for the acceptance rule
    int blockLengthLimit;
    int rleCurrentValue = -1;
    int rleLength = 0;
    uint32_t crc = 0;

    void countRun(int runLength) { blockLength += runLength < 4 ? runLength : 5; }    // :161-190

  public:
    // blockSize = declared block size in bytes (multiple of 100000, :309-320)
    BZip2BlockCompressor(BZip2BitOutputStream &bitOut, int blockSize)
        : bitOutputStream(bitOut), level(blockSize / 100000), blockLengthLimit(blockSize - 6) {
        if (blockSize % 100000 != 0 || level < 1 || level > 9)
            throw IllegalArgumentException("Invalid BZip2 block size " + std::to_string(blockSize));
    }
    bool write(int value) {                                                           // :198-227
        if (blockLength > blockLengthLimit) return false;
        value &= 0xff;
        if (rleLength == 0) {
            rleCurrentValue = value;
            rleLength = 1;
        } else if (rleCurrentValue != value) {
            countRun(rleLength);
            rleCurrentValue = value;
            rleLength = 1;
        } else if (rleLength == 254) {
            countRun(255);
            rleLength = 0;
        } else {
            rleLength++;
        }
        raw.push_back((uint8_t)value);
        return true;
    }
    int write(const uint8_t *data, int offset, int length) {                          // :237-252
        int written = 0;
        while (length-- > 0) {
            if (!write(data[offset++])) break;
            written++;
        }
        return written;
    }
    bool isEmpty() const { return blockLength == 0 && rleLength == 0; }               // :296-298
    uint32_t getCRC() const { return crc; }                                           // :304-306
    void close() {                                                                    // :258-289
        if (raw.empty()) return;
        std::vector<uint8_t> bits(bz2hip_compress_bound(raw.size()));
        uint64_t nbits = 0;
        uint32_t crcs[4] = {0, 0, 0, 0}, nblocks = 0;
        int rc = bz2hip_compress_blocks(ctx.h, raw.data(), raw.size(), level, bits.data(), bits.size(),
                                        &nbits, crcs, 4, &nblocks);
        if (rc != BZ2HIP_OK || nblocks != 1)
            throw BZip2Exception(BZip2Exception::IOException,
                                 std::string("bz2hip: ") + bz2hip_last_error(ctx.h));
        crc = crcs[0];
        uint64_t full = nbits >> 3;
        for (uint64_t i = 0; i < full; i++) bitOutputStream.writeBits(8, bits[i]);
        int rem = (int)(nbits & 7);
        if (rem) bitOutputStream.writeBits(rem, (uint32_t)(bits[full] >> (8 - rem)));
    }
};

// ---- S/BZip2OutputStream.swift:31-201
class BZip2OutputStream : public java::io::OutputStream {
    java::io::OutputStream &outputStream;
    detail::Ctx ctx;
    bool streamFinished = false;

    void check(int rc) {
        if (rc == BZ2HIP_E_STATE)
            throw BZip2Exception(BZip2Exception::IOException, "Write beyond end of stream");
        if (rc != BZ2HIP_OK)
            throw BZip2Exception(BZip2Exception::IOException,
                                 std::string("bz2hip: ") + bz2hip_last_error(ctx.h));
    }
    void drain() { // bytes in stream order, exactly what BZip2BitOutputStream would have emitted
        uint8_t buf[1 << 16];
        size_t got = 0;
        do {
            check(bz2hip_stream_read(ctx.h, buf, sizeof(buf), &got));
            for (size_t i = 0; i < got; i++) outputStream.write(buf[i]);
        } while (got > 0);
    }

  public:
    BZip2OutputStream(java::io::OutputStream &out, int blockSizeMultiplier) : outputStream(out) { // :171-189
        if (blockSizeMultiplier < 1 || blockSizeMultiplier > 9)
            throw IllegalArgumentException("Invalid BZip2 block size " + std::to_string(blockSizeMultiplier));
        check(bz2hip_stream_begin(ctx.h, blockSizeMultiplier));
        drain(); // the 4 header bytes reach the sink at construction (:181-184)
    }
    explicit BZip2OutputStream(java::io::OutputStream &out) : BZip2OutputStream(out, 9) {}       // :197-199
    void write(int value) override {                                                              // :67-79
        uint8_t b = (uint8_t)(value & 0xff);
        write(&b, 0, 1);
    }
    void write(const uint8_t *data, int offset, int length) {                                     // :85-104
        if (streamFinished) throw BZip2Exception(BZip2Exception::IOException, "Write beyond end of stream");
        check(bz2hip_stream_write(ctx.h, data + offset, (size_t)length));
        if (bz2hip_stream_pending(ctx.h)) drain();
    }
    void finish() {                                                                               // :149-161
        if (!streamFinished) {
            streamFinished = true;
            check(bz2hip_stream_finish(ctx.h));
            drain();
            outputStream.flush();
        }
    }
    void close() override {                                                                       // :110-113
        finish();
        outputStream.close();
    }
};

// ---- S/BZip2InputStream.swift:45-258 (and BZip2BlockDecompressor, BZip2HuffmanStageDecoder,
// BZip2BitInputStream under it).  The GPU decodes all blocks of the stream at once, so the source
// is read to its end at the first read(); after that bytes are served in order, and an error of
// the stream is thrown where the reference would throw it: after the bytes it would have
// delivered first (a block's bytes precede its CRC check, :209-212).
class BZip2InputStream : public java::io::InputStream {
    java::io::InputStream *inputStream;
    bool headerless;
    bool decodedAll = false;
    std::vector<uint8_t> decoded;
    size_t pos = 0;
    int pending = BZ2HIP_OK;
    std::string pendingText;

    static const char *text(int rc) {
        switch (rc) {
        case BZ2HIP_E_INSUFFICIENT: return "Insufficient data";                        // BitInputStream :63
        case BZ2HIP_E_HEADER: return "Invalid BZip2 header";                           // :177
        case BZ2HIP_E_FORMAT: return "BZip2 stream format error";                      // :240
        case BZ2HIP_E_BLOCK_CRC: return "BZip2 block CRC error";                       // BlockDecompressor :456
        case BZ2HIP_E_STREAM_CRC: return "BZip2 stream CRC error";                     // :233
        case BZ2HIP_E_TABLES: return "BZip2 block Huffman tables invalid";             // BlockDecompressor :235
        case BZ2HIP_E_BLOCK_SIZE: return "BZip2 block exceeds declared block size";    // :288,:307
        case BZ2HIP_E_START_PTR: return "BZip2 start pointer invalid";                 // :333
        case BZ2HIP_E_DECODE: return "Error decoding BZip2 block";                     // HuffmanStageDecoder :158
        default: return nullptr;
        }
    }
    void decodeAll() {
        if (!inputStream) throw BZip2Exception(BZip2Exception::IOException, "Stream closed"); // :157-159
        decodedAll = true;
        std::vector<uint8_t> z;
        if (headerless) { z.push_back('B'); z.push_back('Z'); }                               // :168
        for (int b; (b = inputStream->read()) >= 0;) z.push_back((uint8_t)b);
        detail::Ctx ctx;
        size_t cap = z.size() * 8 + (1u << 20), got = 0;
        // no stream expands by more than ~1.24 million (a block of 255-runs: 45.9 MB from ~37 bytes)
        const size_t cap_max = z.size() * (size_t)1250000 + (1u << 20);
        for (;;) { // BZ2HIP_E_CAP: a block did not fit, try again with more room (bounded)
            try {
                decoded.assign(cap, 0);
            } catch (const std::bad_alloc &) {
                throw BZip2Exception(BZip2Exception::IOException, "bz2hip: out of memory for the decoded stream");
            }
            pending = bz2hip_decompress(ctx.h, z.data(), z.size(), decoded.data(), cap, &got);
            if (pending != BZ2HIP_E_CAP || cap >= cap_max) break;
            cap = std::min(cap * 4, cap_max);
        }
        decoded.resize(got);
        const char *t = text(pending);
        pendingText = t ? t : std::string("bz2hip: ") + bz2hip_last_error(ctx.h);
    }
    void raise() {
        const int rc = pending;
        pending = BZ2HIP_OK; // the reference's streamComplete: later reads return -1
        throw BZip2Exception(rc == BZ2HIP_E_START_PTR ? BZip2Exception::BZip2ExceptionKind
                                                      : BZip2Exception::IOException, pendingText);
    }

  public:
    BZip2InputStream(java::io::InputStream &in, bool headerless_)                              // :250-256
        : inputStream(&in), headerless(headerless_) {}
    int read() override {                                                                       // :91-108
        if (!decodedAll) decodeAll();
        if (pos < decoded.size()) return decoded[pos++];
        if (pending != BZ2HIP_OK) raise();
        return -1;
    }
    int read(uint8_t *destination, int offset, int length) {                                    // :113-128
        if (!decodedAll) decodeAll();
        if (pos == decoded.size()) {
            if (pending != BZ2HIP_OK) raise();
            return -1;
        }
        const size_t take = std::min<size_t>((size_t)length, decoded.size() - pos);
        std::copy(decoded.begin() + pos, decoded.begin() + pos + take, destination + offset);
        pos += take;
        return (int)take;
    }
    void close() override {                                                                     // :134-147
        decoded.clear();
        pos = 0;
        pending = BZ2HIP_OK;
        if (inputStream) {
            java::io::InputStream *s = inputStream;
            inputStream = nullptr;
            decodedAll = false;
            s->close();
        }
    }
};

} // namespace bzip2javapi
