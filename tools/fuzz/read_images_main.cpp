#include "pth_texture_image.h"
#include <cstdio>
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; i++) {
        pth::RgbImage img; std::string err;
        if (pth::read_image_file(argv[i], &img, &err)) ok++; else bad++;
    }
    std::printf("ok %d bad %d\n", ok, bad);
    return 0;
}
