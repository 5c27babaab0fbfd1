"""bzip2-like front end over libbz2hip (the reference has no CLI, README.md:10-14).

    python -m bzip2_amd [-1..-9] [-d] [-c] [-k] [-f] [file ...]

Compresses FILE to FILE.bz2 (or decompresses FILE.bz2 to FILE with -d) on GPU 0; with no file or
with -c it reads stdin / writes stdout.  Decompression continues over concatenated streams
(`cat a.bz2 b.bz2`), as the bzip2 tool does, through bz2hip_decompress_streams (one call per run of
streams of one level).  Needs the built library and a GPU: no CPU fallback."""
import argparse
import os
import sys

from . import Context


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m bzip2_amd", add_help=True)
    for lv in range(1, 10):
        ap.add_argument(f"-{lv}", dest="level", action="store_const", const=lv)
    ap.add_argument("-d", "--decompress", action="store_true")
    ap.add_argument("-c", "--stdout", action="store_true")
    ap.add_argument("-k", "--keep", action="store_true")
    ap.add_argument("-f", "--force", action="store_true")
    ap.add_argument("files", nargs="*")
    a = ap.parse_args(argv)
    level = a.level or 9
    ctx = Context()
    rc = 0

    def compress_stream(fin, fout):
        """fin -> fout through the incremental API (the reference's BZip2OutputStream.write into a
        sink): constant memory, and the library compresses while this thread reads and writes"""
        ctx.stream_begin(level)
        while True:
            piece = fin.read(16 << 20)
            if not piece:
                break
            ctx.stream_write(piece)
            while ctx.stream_pending():
                fout.write(ctx.stream_read(1 << 24))
        ctx.stream_finish()
        while ctx.stream_pending():
            fout.write(ctx.stream_read(1 << 24))

    def decode(z):
        """every stream of z, concatenated -> (code, bytes)"""
        out = b""
        while True:
            code, part = decode_run(z)
            out += part
            used = ctx.decode_consumed() if code == 0 else 0
            if code != 0 or used == 0 or used >= len(z) or z[used:used + 3] != b"BZh":
                return code, out  # trailing bytes that are not a stream are ignored, as bzip2 does
            z = z[used:]

    def decode_run(z):
        """the streams at the front of z that share the first one's level"""
        cap = max(len(z) * 6, 1 << 20)
        # no stream expands by more than ~1.24 million (a block of 255-runs: 45.9 MB from ~37 bytes)
        cap_max = len(z) * 1250000 + (1 << 20)
        while True:
            code, out, _ = ctx.decompress_streams(z, cap)
            if code != -4 or cap >= cap_max:  # BZ2HIP_E_CAP: a block did not fit
                return code, out
            cap = min(cap * 4, cap_max)

    try:
        if not a.files:
            if not a.decompress:
                compress_stream(sys.stdin.buffer, sys.stdout.buffer)
                return rc
            data = sys.stdin.buffer.read()
            if a.decompress:
                code, out = decode(data)
                sys.stdout.buffer.write(out)
                if code:
                    print(f"bzip2_amd: (stdin): error {code}", file=sys.stderr)
                    rc = 2
            return rc
        for path in a.files:
            if not a.decompress:
                dst = path + ".bz2"
                if a.stdout:
                    with open(path, "rb") as f:
                        compress_stream(f, sys.stdout.buffer)
                    continue
                if os.path.exists(dst) and not a.force:
                    print(f"bzip2_amd: {dst} exists (use -f)", file=sys.stderr)
                    rc = 1
                    continue
                with open(path, "rb") as f, open(dst, "wb") as g:
                    compress_stream(f, g)
                if not a.keep:
                    os.remove(path)
                continue
            with open(path, "rb") as f:
                data = f.read()
            if a.decompress:
                code, out = decode(data)
                if code:
                    print(f"bzip2_amd: {path}: error {code} after {len(out)} bytes", file=sys.stderr)
                    rc = 2
                    continue
                dst = path[:-4] if path.endswith(".bz2") else path + ".out"
            if a.stdout:
                sys.stdout.buffer.write(out)
                continue
            if os.path.exists(dst) and not a.force:
                print(f"bzip2_amd: {dst} exists (use -f)", file=sys.stderr)
                rc = 1
                continue
            with open(dst, "wb") as f:
                f.write(out)
            if not a.keep:
                os.remove(path)
    finally:
        ctx.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
