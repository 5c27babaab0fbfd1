// javapi_driver.cpp -- drives include/bzip2javapi.hpp the way user code drives the reference
// classes; tests/test_gpu_javapi.py compares its output files with the oracle.
//   javapi_driver stream   <level> <in> <out>   BZip2OutputStream, ragged write(data,off,len), close()
//   javapi_driver bytewise <level> <in> <out>   BZip2OutputStream.write(int) per byte, finish()
//   javapi_driver block    <level> <in> <out>   BZip2BlockCompressor + BZip2BitOutputStream, the loop of
//                                               S/BZip2OutputStream.swift:85-104,131-161 written by hand
//   javapi_driver decode|decode1 0 <in.bz2> <out>  BZip2InputStream (block reads | byte reads)
//   javapi_driver errors                        error behaviour; prints one line per check
#include "../../include/bzip2javapi.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>

using namespace bzip2javapi;

static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const char *p, const std::vector<uint8_t> &v) {
    std::ofstream f(p, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)v.size());
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::string mode = argv[1];
    try {
        if (mode == "errors") {
            java::io::ByteArrayOutputStream sink;
            for (int lvl : {0, 10}) {
                try {
                    BZip2OutputStream s(sink, lvl);
                    std::printf("level %d: no exception\n", lvl);
                } catch (const IllegalArgumentException &e) {
                    std::printf("level %d: IllegalArgumentException %s\n", lvl, e.what());
                }
            }
            BZip2OutputStream s(sink, 9);
            std::printf("header bytes at construction: %zu\n", sink.buf.size());
            s.write('a');
            s.finish();
            s.finish(); // idempotent
            try {
                s.write('b');
                std::printf("write after finish: no exception\n");
            } catch (const BZip2Exception &e) {
                std::printf("write after finish: IOException %s\n", e.what());
            }
            s.close();
            std::printf("sink closed: %d\n", sink.closed ? 1 : 0);
            return 0;
        }
        if (mode == "decode" || mode == "decode1") {
            // javapi_driver decode <ignored> <in.bz2> <out>: BZip2InputStream as user code would use it;
            // an exception is reported after the bytes delivered before it
            if (argc < 5) return 2;
            java::io::ByteArrayInputStream src(slurp(argv[3]));
            BZip2InputStream s(src, false);
            std::vector<uint8_t> got;
            try {
                if (mode == "decode1") {
                    for (int b; (b = s.read()) >= 0;) got.push_back((uint8_t)b);
                } else {
                    uint8_t buf[70000];
                    for (int k; (k = s.read(buf, 0, (int)sizeof(buf))) >= 0;) got.insert(got.end(), buf, buf + k);
                }
                std::printf("ok %zu\n", got.size());
            } catch (const BZip2Exception &e) {
                std::printf("%s %s after %zu\n", e.kind == BZip2Exception::IOException ? "IOException" : "BZip2Exception",
                            e.what(), got.size());
            }
            std::printf("after end: %d\n", s.read());
            s.close();
            std::printf("source closed: %d\n", src.closed ? 1 : 0);
            try {
                s.read();
                std::printf("read after close: no exception\n");
            } catch (const BZip2Exception &e) {
                std::printf("read after close: IOException %s\n", e.what());
            }
            spit(argv[4], got);
            return 0;
        }
        if (argc < 5) return 2;
        int level = std::atoi(argv[2]);
        std::vector<uint8_t> in = slurp(argv[3]);
        java::io::ByteArrayOutputStream sink;
        if (mode == "stream") {
            BZip2OutputStream s(sink, level);
            size_t pos = 0, step = 1;
            while (pos < in.size()) {
                size_t k = std::min(in.size() - pos, step);
                s.write(in.data(), (int)pos, (int)k);
                pos += k;
                step = step * 3 + 7;
                if (step > (1u << 22)) step = 1;
            }
            s.close();
        } else if (mode == "bytewise") {
            BZip2OutputStream s(sink, level);
            for (uint8_t b : in) s.write((int)b);
            s.finish();
        } else if (mode == "block") {
            // S/BZip2OutputStream.swift:171-189
            BZip2BitOutputStream bitOut(sink);
            bitOut.writeBits(16, 0x425a);
            bitOut.writeBits(8, 0x68);
            bitOut.writeBits(8, (uint32_t)('0' + level));
            uint32_t streamCRC = 0;
            BZip2BlockCompressor *bc = new BZip2BlockCompressor(bitOut, level * 100000);
            auto closeBlock = [&]() { // :131-141
                if (bc->isEmpty()) return;
                bc->close();
                uint32_t blockCRC = bc->getCRC();
                streamCRC = ((streamCRC << 1) | (streamCRC >> 31)) ^ blockCRC;
            };
            int offset = 0, length = (int)in.size();
            while (length > 0) { // :94-102
                int w = bc->write(in.data(), offset, length);
                if (w < length) {
                    closeBlock();
                    delete bc;
                    bc = new BZip2BlockCompressor(bitOut, level * 100000);
                }
                offset += w;
                length -= w;
            }
            closeBlock(); // :149-161
            delete bc;
            bitOut.writeBits(24, 0x177245);
            bitOut.writeBits(24, 0x385090);
            bitOut.writeInteger(streamCRC);
            bitOut.flush();
        } else {
            return 2;
        }
        spit(argv[4], sink.buf);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
}
