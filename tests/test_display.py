"""Live display streaming (the reference's --display-server: src/displays/tev/display.rs, tev_display.rs): the two IPC packets
against a restatement of IPCGen written from the file format, and a whole create + update conversation with a local TCP server
standing in for tev."""
import ctypes as C
import socket
import struct
import threading

import numpy as np

from helpers import pkg

capi = pkg.capi


def _create_packet(name, w, h):               # display.rs:147-174
    b = bytes([4, 1]) + name.encode() + b"\0" + struct.pack("<III", w, h, 3) + b"R\0G\0B\0"
    return struct.pack("<I", len(b) + 4) + b


def _update_packet(name, x, y, w, h, rgb):    # display.rs:176-235
    b = bytes([6, 0]) + name.encode() + b"\0" + struct.pack("<I", 3) + b"R\0G\0B\0" + struct.pack("<IIII", x, y, w, h)
    b += struct.pack("<qqq", 0, 1, 2) + struct.pack("<qqq", 3, 3, 3) + np.ascontiguousarray(rgb, "<f4").tobytes()
    return struct.pack("<I", len(b) + 4) + b


def _lib():
    lib = capi.load_library()
    lib.pth_tev_create_packet.restype = C.c_size_t
    lib.pth_tev_create_packet.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
    lib.pth_tev_update_packet.restype = C.c_size_t
    lib.pth_tev_update_packet.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.pth_display_connect.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    lib.pth_display_start.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32]
    lib.pth_display_update.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pth_display_close.argtypes = [C.c_void_p]
    lib.pth_display_close.restype = None
    return lib


def test_packets_are_ipcgen_byte_for_byte():
    lib = _lib()
    buf = (C.c_ubyte * 4096)()
    n = lib.pth_tev_create_packet(b"out.exr", 640, 480, buf, 4096)
    assert bytes(buf[:n]) == _create_packet("out.exr", 640, 480)
    rgb = np.random.default_rng(3).random((5, 7, 3), dtype=np.float32)
    big = (C.c_ubyte * 8192)()
    n = lib.pth_tev_update_packet(b"out.exr", 11, 22, 7, 5, rgb.ctypes.data_as(C.c_void_p), big, 8192)
    assert bytes(big[:n]) == _update_packet("out.exr", 11, 22, 7, 5, rgb)


def test_conversation_with_a_display_server():
    """Film::render_start, then one update of a 300 x 200 block at (10, 20): gen_tiles cuts it into 128 x 128 pieces, rows first."""
    srv = socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    port = srv.getsockname()[1]
    got = []

    def serve():
        conn, _ = srv.accept()
        data = b""
        while True:
            chunk = conn.recv(1 << 20)
            if not chunk:
                break
            data += chunk
        conn.close()
        got.append(data)

    th = threading.Thread(target=serve)
    th.start()
    lib = _lib()
    d = C.c_void_p()
    err = C.create_string_buffer(256)
    assert lib.pth_display_connect(("127.0.0.1:%d" % port).encode(), C.byref(d), err, 256) == 0, err.value
    assert lib.pth_display_start(d, b"cornell.exr", 512, 512) == 0
    img = np.random.default_rng(5).random((200, 300, 3), dtype=np.float32)
    assert lib.pth_display_update(d, 10, 20, 300, 200, img.ctypes.data_as(C.c_void_p)) == 0
    lib.pth_display_close(d)
    th.join(10)
    srv.close()
    want = _create_packet("cornell.exr", 512, 512)
    for y0 in range(0, 200, 128):
        for x0 in range(0, 300, 128):
            nw, nh = min(128, 300 - x0), min(128, 200 - y0)
            want += _update_packet("cornell.exr", 10 + x0, 20 + y0, nw, nh, img[y0:y0 + nh, x0:x0 + nw])
    assert got and got[0] == want
    # an address without a port is the reference's own error (display.rs:61-66)
    assert lib.pth_display_connect(b"localhost", C.byref(d), err, 256) != 0 and b"host:port" in err.value


import os
import subprocess

import pytest


@pytest.mark.gpu
def test_cli_streams_bands_to_the_display_server(tmp_path):
    """`pbrt_gpu --display-server host:port`: CreateImage at the full resolution, then one UpdateImage per 128-pixel band and
    128-pixel column, whose pixels are the written image's."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "pbrt-r3_amd", "csrc", "pbrt_gpu")
    scene = os.path.join(root, "tests", "scenes", "cornell.pbrt")
    srv = socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    port = srv.getsockname()[1]
    got = []

    def serve():
        conn, _ = srv.accept()
        data = b""
        while True:
            chunk = conn.recv(1 << 20)
            if not chunk:
                break
            data += chunk
        conn.close()
        got.append(data)

    th = threading.Thread(target=serve)
    th.start()
    out = str(tmp_path / "c.pfm")
    subprocess.check_call([exe, scene, "-o", out, "--pixelsamples", "4", "--quiet", "--display-server", "127.0.0.1:%d" % port])
    th.join(20)
    srv.close()
    data = got[0]
    packets = []
    off = 0
    while off < len(data):
        (n,) = struct.unpack_from("<I", data, off)
        packets.append(data[off:off + n])
        off += n
    assert packets[0][4] == 4 and all(p[4] == 6 for p in packets[1:]) and len(packets) > 1
    # the last band's pixels equal the final image's rows (PFM is stored bottom row first)
    with open(out, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        f.readline()
        img = np.frombuffer(f.read(), "<f4").reshape(h, w, 3)[::-1]
    last = packets[-1]
    name_end = last.index(b"\0", 6)
    x, y, pw, ph = struct.unpack_from("<IIII", last, name_end + 1 + 4 + 6)
    px = np.frombuffer(last[-pw * ph * 12:], "<f4").reshape(ph, pw, 3)
    assert np.array_equal(px.view(np.uint32), np.ascontiguousarray(img[y:y + ph, x:x + pw]).view(np.uint32))
