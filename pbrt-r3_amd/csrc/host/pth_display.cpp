// pth_display.cpp -- live display streaming to a tev image viewer (the reference's --display-server: src/displays/tev/display.rs,
// tev_display.rs, fed by Film::update_display, src/core/film/film.rs:278-360, :424-438).  tev's IPC is a TCP stream of packets
// {u32 length (itself included), u8 directive, payload}; the two directives used are CreateImage (4) and UpdateImage (6), laid out
// exactly as IPCGen::create_image / IPCGen::update_image write them (display.rs:147-235); updates are cut into 128 x 128 tiles
// (DisplayItem::gen_tiles, display.rs:266-345).
#include <arpa/inet.h>
#include <netdb.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/pbrtgpu_host.h"

namespace {
void put(std::vector<unsigned char>& b, const void* p, size_t n) { const unsigned char* q = (const unsigned char*)p; b.insert(b.end(), q, q + n); }
void put_u32(std::vector<unsigned char>& b, uint32_t v) { put(b, &v, 4); }      // little endian hosts only (x86-64)
void put_i64(std::vector<unsigned char>& b, int64_t v) { put(b, &v, 8); }
void put_str(std::vector<unsigned char>& b, const char* s) { put(b, s, std::strlen(s) + 1); }
const char* const kChannels[3] = {"R", "G", "B"};

std::vector<unsigned char> create_packet(const char* name, uint32_t w, uint32_t h) {       // display.rs:147-174
    std::vector<unsigned char> b;
    put_u32(b, 0);
    b.push_back(4);           // DisplayDirective::CreateImage
    b.push_back(1);           // grab_focus
    put_str(b, name);
    put_u32(b, w); put_u32(b, h); put_u32(b, 3);
    for (const char* c : kChannels) put_str(b, c);
    uint32_t len = (uint32_t)b.size();
    std::memcpy(b.data(), &len, 4);
    return b;
}
std::vector<unsigned char> update_packet(const char* name, uint32_t x, uint32_t y, uint32_t w, uint32_t h, const float* rgb) {   // display.rs:176-235
    std::vector<unsigned char> b;
    put_u32(b, 0);
    b.push_back(6);           // DisplayDirective::UpdateImage
    b.push_back(0);           // grab_focus
    put_str(b, name);
    put_u32(b, 3);
    for (const char* c : kChannels) put_str(b, c);
    put_u32(b, x); put_u32(b, y); put_u32(b, w); put_u32(b, h);
    for (int i = 0; i < 3; i++) put_i64(b, i);          // channel offsets
    for (int i = 0; i < 3; i++) put_i64(b, 3);          // channel strides
    put(b, rgb, (size_t)w * h * 3 * sizeof(float));
    uint32_t len = (uint32_t)b.size();
    std::memcpy(b.data(), &len, 4);
    return b;
}
}  // namespace

struct pth_display {
    int fd = -1;
    std::string title;
};

extern "C" {

size_t pth_tev_create_packet(const char* name, uint32_t width, uint32_t height, unsigned char* out, size_t cap) {
    std::vector<unsigned char> b = create_packet(name, width, height);
    if (out && cap >= b.size()) std::memcpy(out, b.data(), b.size());
    return b.size();
}
size_t pth_tev_update_packet(const char* name, uint32_t x, uint32_t y, uint32_t width, uint32_t height, const float* rgb, unsigned char* out, size_t cap) {
    std::vector<unsigned char> b = update_packet(name, x, y, width, height, rgb);
    if (out && cap >= b.size()) std::memcpy(out, b.data(), b.size());
    return b.size();
}

pt_status pth_display_connect(const char* host_port, pth_display** out, char* err, size_t err_cap) {
    auto fail = [&](const std::string& m) { if (err && err_cap) std::snprintf(err, err_cap, "%s", m.c_str()); return PT_ERR_INVALID_ARGUMENT; };
    if (!host_port || !out) return PT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    std::string hp = host_port;
    size_t colon = hp.find_last_of(':');
    if (colon == std::string::npos) return fail("Expected \"host:port\" for display server address. Given \"" + hp + "\".");     // display.rs:61-66
    std::string host = hp.substr(0, colon), port = hp.substr(colon + 1);
    addrinfo hints;
    std::memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_INET;             // the reference keeps the first IPv4 address (display.rs:38)
    hints.ai_socktype = SOCK_STREAM;
    addrinfo* res = nullptr;
    if (getaddrinfo(host.c_str(), port.c_str(), &hints, &res) != 0 || !res) return fail("display server \"" + hp + "\": address not found");
    int fd = socket(res->ai_family, res->ai_socktype, res->ai_protocol);
    if (fd < 0 || connect(fd, res->ai_addr, res->ai_addrlen) != 0) {
        if (fd >= 0) close(fd);
        freeaddrinfo(res);
        return fail("display server \"" + hp + "\": connection failed");
    }
    freeaddrinfo(res);
    pth_display* d = new pth_display;
    d->fd = fd;
    *out = d;
    return PT_OK;
}
static bool send_all(int fd, const std::vector<unsigned char>& b) {
    size_t off = 0;
    while (off < b.size()) {
        ssize_t n = send(fd, b.data() + off, b.size() - off, MSG_NOSIGNAL);
        if (n <= 0) return false;
        off += (size_t)n;
    }
    return true;
}
// Film::render_start (film.rs:424-433): the image is created at the film's FULL resolution, channels R, G, B
pt_status pth_display_start(pth_display* d, const char* title, uint32_t full_width, uint32_t full_height) {
    if (!d || !title) return PT_ERR_INVALID_ARGUMENT;
    d->title = title;
    return send_all(d->fd, create_packet(title, full_width, full_height)) ? PT_OK : PT_ERR_DEVICE;
}
// Film::update_display -> TevDisplay::update -> DisplayItem::update_image: a w x h block of linear RGB whose top-left pixel is (x, y),
// sent as 128 x 128 tiles, rows first
pt_status pth_display_update(pth_display* d, uint32_t x, uint32_t y, uint32_t width, uint32_t height, const float* rgb) {
    if (!d || !rgb) return PT_ERR_INVALID_ARGUMENT;
    const uint32_t T = 128;
    std::vector<float> tile;
    for (uint32_t y0 = 0; y0 < height; y0 += T)
        for (uint32_t x0 = 0; x0 < width; x0 += T) {
            const uint32_t nw = std::min(T, width - x0), nh = std::min(T, height - y0);
            tile.resize((size_t)nw * nh * 3);
            for (uint32_t j = 0; j < nh; j++)
                std::memcpy(&tile[(size_t)j * nw * 3], &rgb[((size_t)(y0 + j) * width + x0) * 3], (size_t)nw * 3 * sizeof(float));
            if (!send_all(d->fd, update_packet(d->title.c_str(), x + x0, y + y0, nw, nh, tile.data()))) return PT_ERR_DEVICE;
        }
    return PT_OK;
}
void pth_display_close(pth_display* d) {
    if (!d) return;
    if (d->fd >= 0) { shutdown(d->fd, SHUT_RDWR); close(d->fd); }
    delete d;
}

}  // extern "C"
