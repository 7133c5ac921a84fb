// hdr_io.hpp — Radiance RGBE (.hdr) pictures for the host harness, in the two places the reference touches the format:
//   * the environment map: `stbi_loadf("env.hdr", &w, &h, &channels, 4)` (Application.cpp:225) -> Sky::hdri_data, RGBA f32, rows top-down,
//     alpha 1, no flip, no gamma (Primitives.hpp:29-47 indexes it as data[4 * (y * width + x)]);
//   * the F5 screenshot: `stbi_flip_vertically_on_write(true); stbi_write_hdr(path, w, h, 4, RGBA_float_data)` (Image.cpp:71-74) of
//     Renderer::GetFrame() — the framebuffer's row 0 is the bottom of the picture (Application.cpp:381), hence the flip.
// stb_image / stb_image_write are third-party single-header libraries the reference includes but does not vendor (Image.cpp:3-10; no
// version pinned, absent from /root/reference and from this image).  What is restated here is their published RGBE arithmetic:
//   decode (stbi__hdr_convert):    e == 0 -> 0, else  channel = byte * 2^(e - 136)   (no +0.5), alpha = 1 for 4 components;
//   encode (stbiw__linear_to_rgbe): m = max(r, g, b); m < 1e-32 -> 0 0 0 0, else frexp(m) = f * 2^e, byte = (uint8)(channel * f * 256 / m), E = e + 128.
// Files: "#?RADIANCE" / "#?RGBE" header lines up to an empty line, "FORMAT=32-bit_rle_rgbe", resolution "-Y H +X W" (the only orientation
// stb reads or writes), then per scanline either flat RGBE quadruples or the "new" run-length form 02 02 hi lo + four byte planes.
// The writer emits the run-length form for 8 <= W < 32768 like stb does (its choice of runs is an encoder detail, any decoder gives the same
// floats), flat quadruples otherwise.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace mirt_hdr {

inline void rgbe_to_float(const uint8_t in[4], float out[4]) {
	if (in[3] != 0) {
		const float f = std::ldexp(1.0f, static_cast<int>(in[3]) - (128 + 8));
		out[0] = in[0] * f; out[1] = in[1] * f; out[2] = in[2] * f;
	} else out[0] = out[1] = out[2] = 0.0f;
	out[3] = 1.0f;
}
inline void float_to_rgbe(const float lin[3], uint8_t out[4]) {
	const float m = lin[0] > (lin[1] > lin[2] ? lin[1] : lin[2]) ? lin[0] : (lin[1] > lin[2] ? lin[1] : lin[2]);
	if (m < 1e-32f) { out[0] = out[1] = out[2] = out[3] = 0; return; }
	int e;
	const float norm = static_cast<float>(std::frexp(m, &e)) * 256.0f / m;
	out[0] = static_cast<uint8_t>(lin[0] * norm); out[1] = static_cast<uint8_t>(lin[1] * norm); out[2] = static_cast<uint8_t>(lin[2] * norm);
	out[3] = static_cast<uint8_t>(e + 128);
}

// Reads a picture as stbi_loadf(path, &w, &h, &n, 4) returns it: RGBA f32, first row = top of the picture.  Returns "" or what went wrong.
inline std::string read(const std::string& path, std::vector<float>& rgba, int32_t& width, int32_t& height) {
	FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) return "cannot open " + path;
	std::vector<uint8_t> bytes;
	{ uint8_t buf[65536]; size_t n; while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n); }
	std::fclose(f);
	size_t at = 0;
	auto line = [&]() { std::string s; while (at < bytes.size() && bytes[at] != '\n') s.push_back(static_cast<char>(bytes[at++])); if (at < bytes.size()) at++; return s; };
	const std::string magic = line();
	if (magic != "#?RADIANCE" && magic != "#?RGBE") return "not a Radiance picture (no #?RADIANCE line)";
	bool format_ok = false;
	for (;;) {
		if (at >= bytes.size()) return "header ends before the resolution line";
		const std::string l = line();
		if (l.empty()) break;
		if (l == "FORMAT=32-bit_rle_rgbe") format_ok = true;
	}
	if (!format_ok) return "unsupported FORMAT (need 32-bit_rle_rgbe)";
	int h = 0, w = 0;
	if (std::sscanf(line().c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24)) return "unsupported resolution line (need -Y H +X W)";
	width = w; height = h;
	rgba.assign(static_cast<size_t>(w) * h * 4, 0.0f);
	std::vector<uint8_t> scan(static_cast<size_t>(w) * 4);
	for (int y = 0; y < h; y++) {
		bool rle = false;
		if (w >= 8 && w < 32768 && at + 4 <= bytes.size() && bytes[at] == 2 && bytes[at + 1] == 2 && !(bytes[at + 2] & 0x80)) {
			if (((bytes[at + 2] << 8) | bytes[at + 3]) != w) return "run-length scanline of the wrong width";
			rle = true; at += 4;
		}
		if (rle) {
			for (int k = 0; k < 4; k++) {
				int x = 0;
				while (x < w) {
					if (at >= bytes.size()) return "file ends inside a scanline";
					int count = bytes[at++];
					if (count > 128) {                                              // a run of count - 128 equal bytes
						count -= 128;
						if (count == 0 || x + count > w || at >= bytes.size()) return "corrupt run";
						const uint8_t v = bytes[at++];
						for (int i = 0; i < count; i++) scan[static_cast<size_t>(x++) * 4 + k] = v;
					} else {                                                        // count literal bytes
						if (count == 0 || x + count > w || at + count > bytes.size()) return "corrupt literal block";
						for (int i = 0; i < count; i++) scan[static_cast<size_t>(x++) * 4 + k] = bytes[at++];
					}
				}
			}
		} else {
			if (at + static_cast<size_t>(w) * 4 > bytes.size()) return "file ends inside a scanline";
			std::memcpy(scan.data(), bytes.data() + at, static_cast<size_t>(w) * 4); at += static_cast<size_t>(w) * 4;
		}
		for (int x = 0; x < w; x++) rgbe_to_float(&scan[static_cast<size_t>(x) * 4], &rgba[(static_cast<size_t>(y) * w + x) * 4]);
	}
	return "";
}

// Image::Store (Image.cpp:71-74): RGBA f32 rows, row 0 = BOTTOM of the picture (Renderer::GetFrame) -> top-down RGBE file, alpha dropped.
inline bool write_flipped(const std::string& path, const float* rgba, uint32_t width, uint32_t height) {
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) return false;
	std::fprintf(f, "#?RADIANCE\n# Written by mirt_headless (Image::Store, Image.cpp:71-74)\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=          1.0000000000000\n\n-Y %u +X %u\n", height, width);
	std::vector<uint8_t> scan(static_cast<size_t>(width) * 4), out;
	for (uint32_t row = 0; row < height; row++) {
		const float* src = rgba + static_cast<size_t>(height - 1 - row) * width * 4;      // stbi_flip_vertically_on_write(true)
		for (uint32_t x = 0; x < width; x++) float_to_rgbe(src + static_cast<size_t>(x) * 4, &scan[static_cast<size_t>(x) * 4]);
		out.clear();
		if (width < 8 || width >= 32768) out.assign(scan.begin(), scan.end());
		else {
			out.push_back(2); out.push_back(2); out.push_back(static_cast<uint8_t>(width >> 8)); out.push_back(static_cast<uint8_t>(width & 0xff));
			for (int k = 0; k < 4; k++) {
				uint32_t x = 0;
				while (x < width) {
					uint32_t run = 1;                                               // equal bytes starting at x
					while (x + run < width && run < 127 && scan[static_cast<size_t>(x + run) * 4 + k] == scan[static_cast<size_t>(x) * 4 + k]) run++;
					if (run >= 3) { out.push_back(static_cast<uint8_t>(128 + run)); out.push_back(scan[static_cast<size_t>(x) * 4 + k]); x += run; continue; }
					uint32_t lit = 0;                                               // literals up to the next run of three
					while (x + lit < width && lit < 128) {
						if (x + lit + 2 < width && scan[static_cast<size_t>(x + lit) * 4 + k] == scan[static_cast<size_t>(x + lit + 1) * 4 + k] && scan[static_cast<size_t>(x + lit) * 4 + k] == scan[static_cast<size_t>(x + lit + 2) * 4 + k]) break;
						lit++;
					}
					out.push_back(static_cast<uint8_t>(lit));
					for (uint32_t i = 0; i < lit; i++) out.push_back(scan[static_cast<size_t>(x + i) * 4 + k]);
					x += lit;
				}
			}
		}
		if (std::fwrite(out.data(), 1, out.size(), f) != out.size()) { std::fclose(f); return false; }
	}
	return std::fclose(f) == 0;
}

} // namespace mirt_hdr
