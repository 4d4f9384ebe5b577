// Input-side file formats, host code only:
//  * RIFF/RIFX 16-bit PCM waveforms: WaveInfo::Read + WaveData::Read (feat/wave-reader.cc:113-318),
//    incl. filler chunks before "fmt ", WAVE_FORMAT_EXTENSIBLE, "fact"/"LIST" chunks before
//    "data" and the streamed sizes SoX writes;
//  * Kaldi float matrix archives ("ark"): entries `key<space>` + object, object = binary
//    ("\0B", token FM / DM, int32 rows, int32 cols, row-major data; matrix/kaldi-matrix.cc:
//    1378-1404, 1470-1512), compressed (CM / CM2 / CM3, matrix/compressed-matrix.cc:566-650) or
//    text (" [\n  1 2 \n  3 4 ]\n", :1405-1417);  int32 vectors (BasicVectorHolder, util/kaldi-holder-inl.h:
//    230-250: "\0B", WriteBasicType(count), WriteBasicType per element; or text "1 2 3 \n").
// This is what lets log-likelihoods computed by Kaldi's nnet3-compute, or features / ivectors
// from its own tools, be fed to the decoder, and audio be read without the reference.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <new>

#include "common.h"

namespace {

struct File {
  FILE *f;
  explicit File(FILE *fp) : f(fp) {}
  ~File() { if (f) fclose(f); }
  bool Bytes(void *p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
  int Get() { return fgetc(f); }
  int Peek() { int c = fgetc(f); if (c != EOF) ungetc(c, f); return c; }
  // can the file still hold `bytes`?  Asked BEFORE allocating for a count read from a header, so that a corrupt one
  // fails here and not in a multi-gigabyte resize (a stream that cannot seek gets the benefit of the doubt up to 2 GB)
  bool Holds(uint64_t bytes) {
    const long here = ftell(f);
    if (here < 0 || fseek(f, 0, SEEK_END) != 0) return bytes <= (1ull << 31);
    const long end = ftell(f);
    (void)fseek(f, here, SEEK_SET);
    return end >= here && bytes <= static_cast<uint64_t>(end - here);
  }
};

int Fail(const char *what, const char *path) { return kamd::SetError(KAMD_ERR_ARG, "%s: %s", what, path ? path : ""); }

inline uint32_t Swap4(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }
inline uint16_t Swap2(uint16_t v) { return static_cast<uint16_t>((v >> 8) | (v << 8)); }

// space-terminated token (base/io-funcs.cc ReadToken, binary mode)
bool ReadToken(File *f, std::string *tok) {
  tok->clear();
  int c;
  while ((c = f->Get()) != EOF && c != ' ') { tok->push_back(static_cast<char>(c)); if (tok->size() > 64) return false; }
  return c == ' ';
}
// ReadBasicType<int32>, binary (base/io-funcs-inl.h:55-90): size byte 4, then the value
bool ReadI32(File *f, int32_t *v) { return f->Get() == 4 && f->Bytes(v, 4); }

bool ReadMatrixBody(File *f, bool binary, std::vector<float> *data, int32_t *rows, int32_t *cols, std::string *err) {
  if (binary) {
    std::string tok;
    if (!ReadToken(f, &tok)) { *err = "matrix token expected"; return false; }
    if (tok == "FM" || tok == "DM") {
      if (!ReadI32(f, rows) || !ReadI32(f, cols) || *rows < 0 || *cols < 0) { *err = "bad matrix size"; return false; }
      const size_t n = static_cast<size_t>(*rows) * *cols;
      if (!f->Holds(static_cast<uint64_t>(n) * (tok == "FM" ? 4 : 8))) { *err = "matrix size exceeds the file size"; return false; }
      data->resize(n);
      if (tok == "FM") { if (!f->Bytes(data->data(), n * 4)) { *err = "truncated matrix"; return false; } }
      else {
        std::vector<double> d(n);
        if (!f->Bytes(d.data(), n * 8)) { *err = "truncated matrix"; return false; }
        for (size_t i = 0; i < n; i++) (*data)[i] = static_cast<float>(d[i]);
      }
      return true;
    }
    if (tok == "FV" || tok == "DV") {                  // Vector<float>::Write (kaldi-vector.cc:1110-1133): one row
      if (!ReadI32(f, cols) || *cols < 0) { *err = "bad vector size"; return false; }
      *rows = 1;
      const size_t n = static_cast<size_t>(*cols);
      if (!f->Holds(static_cast<uint64_t>(n) * (tok == "FV" ? 4 : 8))) { *err = "vector size exceeds the file size"; return false; }
      data->resize(n);
      if (tok == "FV") { if (!f->Bytes(data->data(), n * 4)) { *err = "truncated vector"; return false; } }
      else {
        std::vector<double> d(n);
        if (!f->Bytes(d.data(), n * 8)) { *err = "truncated vector"; return false; }
        for (size_t i = 0; i < n; i++) (*data)[i] = static_cast<float>(d[i]);
      }
      return true;
    }
    if (tok == "CM" || tok == "CM2" || tok == "CM3") {
      // GlobalHeader without its first field (compressed-matrix.cc:576-584)
      struct { float min_value, range; int32_t num_rows, num_cols; } h;
      if (!f->Bytes(&h, sizeof(h)) || h.num_rows < 0 || h.num_cols < 0) { *err = "bad compressed header"; return false; }
      *rows = h.num_rows; *cols = h.num_cols;
      const size_t R = h.num_rows, C = h.num_cols;
      if (!f->Holds(static_cast<uint64_t>(R) * C * (tok == "CM2" ? 2 : 1))) { *err = "compressed matrix size exceeds the file size"; return false; }
      data->assign(R * C, 0.0f);
      if (C == 0) return true;
      auto u16 = [&](uint16_t v) { return h.min_value + h.range * 1.52590218966964e-05F * v; };   // :371-377
      if (tok == "CM") {        // one byte per element, column headers, column-major bytes
        std::vector<uint16_t> ph(4 * C);
        std::vector<uint8_t> b(R * C);
        if (!f->Bytes(ph.data(), ph.size() * 2) || !f->Bytes(b.data(), b.size())) { *err = "truncated compressed matrix"; return false; }
        for (size_t c = 0; c < C; c++) {
          const float p0 = u16(ph[4 * c]), p25 = u16(ph[4 * c + 1]), p75 = u16(ph[4 * c + 2]), p100 = u16(ph[4 * c + 3]);
          for (size_t r = 0; r < R; r++) {
            const uint8_t v = b[c * R + r];
            float x;                                   // CharToFloat (:490-500)
            if (v <= 64) x = p0 + (p25 - p0) * v * (1 / 64.0);
            else if (v <= 192) x = p25 + (p75 - p25) * (v - 64) * (1 / 128.0);
            else x = p75 + (p100 - p75) * (v - 192) * (1 / 63.0);
            (*data)[r * C + c] = x;
          }
        }
      } else if (tok == "CM2") {
        std::vector<uint16_t> b(R * C);
        if (!f->Bytes(b.data(), b.size() * 2)) { *err = "truncated compressed matrix"; return false; }
        for (size_t i = 0; i < R * C; i++) (*data)[i] = u16(b[i]);
      } else {
        std::vector<uint8_t> b(R * C);
        if (!f->Bytes(b.data(), b.size())) { *err = "truncated compressed matrix"; return false; }
        const float inc = h.range * (1.0f / 255.0f);
        for (size_t i = 0; i < R * C; i++) (*data)[i] = h.min_value + b[i] * inc;
      }
      return true;
    }
    *err = "expected token FM, DM, CM*, FV or DV, got " + tok;
    return false;
  }
  // text: [ rows separated by newlines ]
  int c;
  while ((c = f->Get()) != EOF && (c == ' ' || c == '\t' || c == '\n')) {}
  if (c != '[') { *err = "'[' expected"; return false; }
  data->clear(); *rows = 0; *cols = 0;
  std::vector<float> row;
  std::string num;
  auto flush_num = [&]() -> bool {
    if (num.empty()) return true;
    char *e = NULL;
    float v;
    if (num == "inf" || num == "Inf") v = INFINITY; else if (num == "-inf" || num == "-Inf") v = -INFINITY;
    else if (num == "nan" || num == "NaN") v = NAN;
    else { v = strtof(num.c_str(), &e); if (e == num.c_str() || *e) return false; }
    row.push_back(v); num.clear();
    return true;
  };
  auto flush_row = [&]() -> bool {
    if (row.empty()) return true;
    if (*cols == 0) *cols = static_cast<int32_t>(row.size());
    if (static_cast<int32_t>(row.size()) != *cols) return false;
    data->insert(data->end(), row.begin(), row.end());
    (*rows)++; row.clear();
    return true;
  };
  while ((c = f->Get()) != EOF) {
    if (c == ']') {
      if (!flush_num() || !flush_row()) { *err = "bad text matrix"; return false; }
      while ((c = f->Peek()) == ' ' || c == '\r') f->Get();
      if (f->Peek() == '\n') f->Get();
      return true;
    }
    if (c == '\n' || c == ';') { if (!flush_num() || !flush_row()) { *err = "ragged text matrix"; return false; } }
    else if (c == ' ' || c == '\t' || c == '\r') { if (!flush_num()) { *err = "bad number in text matrix"; return false; } }
    else num.push_back(static_cast<char>(c));
  }
  *err = "']' expected";
  return false;
}

}  // namespace

extern "C" {

// WaveData::Read.  *data (malloc'ed, kamd_host_free) holds num_channels rows of num_samples
// floats in int16 range, as Kaldi keeps them (feat/wave-reader.h:60-75).
int kamd_wave_read(const char *path, float *samp_freq, int32_t *num_channels, int64_t *num_samples, float **data) {
  *data = NULL; *samp_freq = 0; *num_channels = 0; *num_samples = 0;
  File f(fopen(path, "rb"));
  if (!f.f) return Fail("cannot open", path);
  char tag[5] = {0, 0, 0, 0, 0};
  bool swap = false;
  auto rtag = [&]() { return f.Bytes(tag, 4); };
  auto ru32 = [&](uint32_t *v) { if (!f.Bytes(v, 4)) return false; if (swap) *v = Swap4(*v); return true; };
  auto ru16 = [&](uint16_t *v) { if (!f.Bytes(v, 2)) return false; if (swap) *v = Swap2(*v); return true; };
  if (!rtag()) return Fail("WaveData: empty file", path);
  if (!strcmp(tag, "RIFF")) swap = false; else if (!strcmp(tag, "RIFX")) swap = true;
  else return kamd::SetError(KAMD_ERR_ARG, "WaveData: expected RIFF or RIFX, got %s: %s", tag, path);
  uint32_t riff_size, sz;
  if (!ru32(&riff_size) || !rtag() || strcmp(tag, "WAVE")) return Fail("WaveData: expected WAVE", path);
  if (!rtag()) return Fail("WaveData: truncated header", path);
  while (strcmp(tag, "fmt ")) {                         // filler chunks (e.g. Apple's JUNK)
    if (!ru32(&sz) || fseek(f.f, sz, SEEK_CUR) != 0 || !rtag()) return Fail("WaveData: fmt chunk not found", path);
  }
  uint32_t fmt_size, sample_rate, byte_rate;
  uint16_t audio_format, nch, block_align, bits;
  if (!ru32(&fmt_size) || !ru16(&audio_format) || !ru16(&nch) || !ru32(&sample_rate) || !ru32(&byte_rate) || !ru16(&block_align) || !ru16(&bits))
    return Fail("WaveData: truncated fmt chunk", path);
  uint32_t fmt_read = 16;
  if (audio_format == 1) {
    if (fmt_size < 16) return Fail("WaveData: expect PCM format data to have fmt chunk of at least size 16", path);
  } else if (audio_format == 0xFFFE) {                  // WAVE_FORMAT_EXTENSIBLE with the PCM sub-format GUID
    uint16_t extra, u; uint32_t mask, g1, g2, g3, g4;
    if (!ru16(&extra) || fmt_size < 40 || extra < 22 || !ru16(&u) || !ru32(&mask) || !ru32(&g1) || !ru32(&g2) || !ru32(&g3) || !ru32(&g4))
      return Fail("WaveData: malformed WAVE_FORMAT_EXTENSIBLE format data", path);
    if (g1 != 0x00000001 || g2 != 0x00100000 || g3 != 0xAA000080 || g4 != 0x719B3800) return Fail("WaveData: unsupported WAVE_FORMAT_EXTENSIBLE format", path);
    fmt_read = 40;
  } else return kamd::SetError(KAMD_ERR_ARG, "WaveData: can read only PCM data, format id in file is: %d: %s", audio_format, path);
  if (fmt_size > fmt_read && fseek(f.f, fmt_size - fmt_read, SEEK_CUR) != 0) return Fail("WaveData: truncated fmt chunk", path);
  if (nch == 0) return Fail("WaveData: no channels present", path);
  if (bits != 16) return kamd::SetError(KAMD_ERR_ARG, "WaveData: unsupported bits_per_sample = %d: %s", bits, path);
  if (byte_rate != sample_rate * 2 * nch) return Fail("WaveData: unexpected byte rate", path);
  if (block_align != nch * 2) return Fail("WaveData: unexpected block_align", path);
  if (!rtag()) return Fail("WaveData: data chunk not found", path);
  while (strcmp(tag, "data")) {                         // "fact", "LIST", ...
    if (!ru32(&sz) || fseek(f.f, sz, SEEK_CUR) != 0 || !rtag()) return Fail("WaveData: data chunk not found", path);
  }
  uint32_t data_size;
  if (!ru32(&data_size)) return Fail("WaveData: truncated data chunk", path);
  const bool streamed = riff_size == 0 || riff_size == 0xFFFFFFFFu || data_size == 0 || data_size == 0xFFFFFFFFu || data_size == 0x7FFFF000u;
  std::vector<char> buf;
  if (streamed) {
    char tmp[65536]; size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f.f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  } else {
    buf.resize(data_size);
    const size_t n = fread(buf.data(), 1, data_size, f.f);
    buf.resize(n);                                      // "Truncated file?" is a warning in the reference
  }
  if (buf.empty()) return Fail("WaveData: empty file (no data)", path);
  const int64_t N = static_cast<int64_t>(buf.size() / block_align);
  float *out = static_cast<float *>(malloc(sizeof(float) * (static_cast<size_t>(N) * nch + 1)));
  if (!out) return kamd::SetError(KAMD_ERR_ARG, "out of host memory");
  const uint16_t *p = reinterpret_cast<const uint16_t *>(buf.data());
  for (int64_t i = 0; i < N; i++)
    for (int c = 0; c < nch; c++) {
      uint16_t v = *p++;
      if (swap) v = Swap2(v);
      out[static_cast<size_t>(c) * N + i] = static_cast<float>(static_cast<int16_t>(v));
    }
  *data = out; *samp_freq = static_cast<float>(sample_rate); *num_channels = nch; *num_samples = N;
  return KAMD_OK;
}

// Next float-matrix entry of an archive at byte *offset (advanced past it).  Returns 1 at end
// of file.  *data is malloc'ed (kamd_host_free), row-major [rows x cols].
int kamd_ark_read_matrix(const char *path, int64_t *offset, char *key, int key_cap, int32_t *rows, int32_t *cols,
                         float **data) {
  *data = NULL; *rows = 0; *cols = 0;
  File f(fopen(path, "rb"));
  if (!f.f) return Fail("cannot open", path);
  if (fseek(f.f, static_cast<long>(*offset), SEEK_SET) != 0) return Fail("seek failed", path);
  int c;
  const char *shown = "(scp entry)";
  if (key) {
    while ((c = f.Get()) != EOF && (c == '\n' || c == ' ')) {}
    if (c == EOF) return 1;
    std::string k(1, static_cast<char>(c));
    while ((c = f.Get()) != EOF && c != ' ' && c != '\n') k.push_back(static_cast<char>(c));
    if (c != ' ' || static_cast<int>(k.size()) + 1 > key_cap) return Fail("bad archive key", path);
    memcpy(key, k.c_str(), k.size() + 1);
    shown = key;
  } else if (f.Peek() == EOF) return Fail("offset past the end", path);
  bool binary = false;
  if (f.Peek() == '\0') { f.Get(); if (f.Get() != 'B') return Fail("bad binary marker", path); binary = true; }
  std::vector<float> m;
  std::string err;
  try {
    if (!ReadMatrixBody(&f, binary, &m, rows, cols, &err)) return kamd::SetError(KAMD_ERR_ARG, "%s: key %s: %s", path, shown, err.c_str());
  } catch (const std::bad_alloc &) {      // no exception may cross the C ABI
    return kamd::SetError(KAMD_ERR_ARG, "%s: key %s: out of host memory reading a %d x %d matrix", path, shown, *rows, *cols);
  }
  *offset = ftell(f.f);
  *data = static_cast<float *>(malloc(sizeof(float) * (m.size() + 1)));
  if (!*data) return kamd::SetError(KAMD_ERR_ARG, "out of host memory");
  if (!m.empty()) memcpy(*data, m.data(), m.size() * sizeof(float));
  return KAMD_OK;
}

// BaseFloatMatrixWriter entry (util/kaldi-holder-inl.h KaldiObjectHolder<Matrix<float>>::Write)
int kamd_ark_write_matrix(const char *path, int append, const char *key, int binary, int32_t rows, int32_t cols,
                          const float *data) {
  FILE *f = fopen(path, append ? "ab" : "wb");
  if (!f) return Fail("cannot open for writing", path);
  fputs(key, f); fputc(' ', f);
  if (binary) {
    fputc('\0', f); fputc('B', f);
    fputs("FM ", f);
    fputc(4, f); fwrite(&rows, 4, 1, f); fputc(4, f); fwrite(&cols, 4, 1, f);
    fwrite(data, sizeof(float), static_cast<size_t>(rows) * cols, f);
  } else if (cols == 0) {
    fputs(" [ ]\n", f);
  } else {
    fputs(" [", f);
    for (int32_t i = 0; i < rows; i++) {
      fputs("\n  ", f);
      for (int32_t j = 0; j < cols; j++) fprintf(f, "%.9g ", data[static_cast<size_t>(i) * cols + j]);
    }
    fputs("]\n", f);
  }
  const bool ok = !ferror(f);
  fclose(f);
  return ok ? KAMD_OK : Fail("write failed", path);
}

// Next int32-vector entry (Int32VectorHolder: alignments, word sequences, a dumped id2pdf table)
int kamd_ark_read_int32_vector(const char *path, int64_t *offset, char *key, int key_cap, int32_t *n, int32_t **data) {
  *data = NULL; *n = 0;
  File f(fopen(path, "rb"));
  if (!f.f) return Fail("cannot open", path);
  if (fseek(f.f, static_cast<long>(*offset), SEEK_SET) != 0) return Fail("seek failed", path);
  int c = ' ';
  const char *shown = "(scp entry)";
  if (key) {
    while ((c = f.Get()) != EOF && (c == '\n' || c == ' ')) {}
    if (c == EOF) return 1;
    std::string k(1, static_cast<char>(c));
    while ((c = f.Get()) != EOF && c != ' ' && c != '\n') k.push_back(static_cast<char>(c));
    if ((c != ' ' && c != '\n') || static_cast<int>(k.size()) + 1 > key_cap) return Fail("bad archive key", path);
    memcpy(key, k.c_str(), k.size() + 1);
    shown = key;
  }
  std::vector<int32_t> v;
  if (c == ' ' && f.Peek() == '\0') {
    f.Get();
    // BasicVectorHolder<int32>::Write (util/kaldi-holder-inl.h:230-250): WriteBasicType(size), then
    // WriteBasicType of every element (each with its own size byte)
    int32_t cnt;
    if (f.Get() != 'B' || !ReadI32(&f, &cnt) || cnt < 0) return Fail("bad binary int32 vector", path);
    if (!f.Holds(static_cast<uint64_t>(cnt) * 5)) return Fail("int32 vector size exceeds the file size", path);
    v.resize(cnt);
    for (int32_t i = 0; i < cnt; i++) if (!ReadI32(&f, &v[i])) return Fail("truncated int32 vector", path);
  } else if (c == ' ') {
    std::string line;
    while ((c = f.Get()) != EOF && c != '\n') line.push_back(static_cast<char>(c));
    const char *p = line.c_str();
    while (*p) {
      while (*p == ' ' || *p == '\t' || *p == '\r') p++;
      if (!*p) break;
      char *e;
      const long x = strtol(p, &e, 10);
      if (e == p) return kamd::SetError(KAMD_ERR_ARG, "%s: key %s: bad integer", path, shown);
      v.push_back(static_cast<int32_t>(x)); p = e;
    }
  }
  *offset = ftell(f.f);
  *n = static_cast<int32_t>(v.size());
  *data = static_cast<int32_t *>(malloc(sizeof(int32_t) * (v.size() + 1)));
  if (!*data) return kamd::SetError(KAMD_ERR_ARG, "out of host memory");
  if (!v.empty()) memcpy(*data, v.data(), v.size() * 4);
  return KAMD_OK;
}

}  // extern "C"
