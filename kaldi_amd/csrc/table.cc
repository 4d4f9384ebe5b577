// Kaldi's "extended filenames" and table specifiers (host code only):
//   ClassifyRxfilename / ClassifyWxfilename   util/kaldi-io.cc:85-186
//   ClassifyRspecifier / ClassifyWspecifier   util/kaldi-table.cc:115-310
//   ReadScriptFile                             util/kaldi-table.cc:56-84
// plus kamd_rx_materialize, which turns any rxfilename ("file", "file:offset", "cmd |", "-") into
// a seekable (path, offset) pair the format readers of kaldi_io.cc / fst_io.cc take: pipe and
// stdin contents are spooled to a temporary file, which is what lets the same readers serve
// `scp:` lines that are commands (the usual wav.scp with sox/sph2pipe) and `ark:gunzip -c ...|`.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

#include "common.h"

namespace {

// SplitStringToVector(str, ", ", false, &out) (util/text-utils.cc): empty fields are kept
std::vector<std::string> SplitKeepEmpty(const std::string &s, const char *delims) {
  std::vector<std::string> out;
  size_t start = 0;
  for (;;) {
    const size_t e = s.find_first_of(delims, start);
    out.push_back(s.substr(start, e == std::string::npos ? std::string::npos : e - start));
    if (e == std::string::npos) break;
    start = e + 1;
  }
  return out;
}

int CopyOut(const std::string &s, char *dst, int cap) {
  if (!dst) return 0;
  if (static_cast<int>(s.size()) + 1 > cap) return kamd::SetError(KAMD_ERR_ARG, "name buffer too small (%zu bytes needed)", s.size() + 1);
  memcpy(dst, s.c_str(), s.size() + 1);
  return 0;
}

int RspecType(const std::string &r, std::string *rx, int *opts) {
  if (rx) rx->clear();
  int o = 0;
  const size_t pos = r.find(':');
  if (pos == std::string::npos) return 0;
  if (isspace(static_cast<unsigned char>(r.back()))) return 0;
  int rs = 0;
  for (const std::string &c : SplitKeepEmpty(r.substr(0, pos), ", ")) {
    if (c == "b" || c == "t") {}
    else if (c == "o") o |= KAMD_RSPEC_ONCE; else if (c == "no") o &= ~KAMD_RSPEC_ONCE;
    else if (c == "p") o |= KAMD_RSPEC_PERMISSIVE; else if (c == "np") o &= ~KAMD_RSPEC_PERMISSIVE;
    else if (c == "s") o |= KAMD_RSPEC_SORTED; else if (c == "ns") o &= ~KAMD_RSPEC_SORTED;
    else if (c == "cs") o |= KAMD_RSPEC_CALLED_SORTED; else if (c == "ncs") o &= ~KAMD_RSPEC_CALLED_SORTED;
    else if (c == "bg") o |= KAMD_RSPEC_BACKGROUND;
    else if (c == "ark") { if (rs) return 0; rs = 1; }
    else if (c == "scp") { if (rs) return 0; rs = 2; }
    else return 0;
  }
  if (rs && rx) *rx = r.substr(pos + 1);
  if (opts) *opts = o;
  return rs;
}

int WspecType(const std::string &w, std::string *ark, std::string *scp, int *opts) {
  if (ark) ark->clear();
  if (scp) scp->clear();
  const size_t pos = w.find(':');
  if (pos == std::string::npos) return 0;
  if (isspace(static_cast<unsigned char>(w.back()))) return 0;
  const std::string after = w.substr(pos + 1);
  int o = KAMD_WSPEC_BINARY, ws = 0;
  for (const std::string &c : SplitKeepEmpty(w.substr(0, pos), ", ")) {
    if (c == "b") o |= KAMD_WSPEC_BINARY; else if (c == "t") o &= ~KAMD_WSPEC_BINARY;
    else if (c == "f") o |= KAMD_WSPEC_FLUSH; else if (c == "nf") o &= ~KAMD_WSPEC_FLUSH;
    else if (c == "p") o |= KAMD_WSPEC_PERMISSIVE;
    else if (c == "ark") { if (ws) return 0; ws = 1; }
    else if (c == "scp") { if (ws == 0) ws = 2; else if (ws == 1) ws = 3; else return 0; }
    else return 0;
  }
  if (ws == 1 && ark) *ark = after;
  if (ws == 2 && scp) *scp = after;
  if (ws == 3) {
    const size_t comma = after.find(',');
    if (comma == std::string::npos) return 0;
    if (ark) *ark = after.substr(0, comma);
    if (scp) *scp = after.substr(comma + 1);
  }
  if (opts) *opts = o;
  return ws;
}

bool LooksLikeSpecifier(const std::string &f) {
  return (f[0] == 'a' || f[0] == 's') && f.find(':') != std::string::npos &&
         (WspecType(f, NULL, NULL, NULL) != 0 || RspecType(f, NULL, NULL) != 0);
}
// "name:123": a colon followed by digits only, up to the end
bool EndsInOffset(const std::string &f) {
  if (f.empty() || !isdigit(static_cast<unsigned char>(f.back()))) return false;
  size_t d = f.size() - 1;
  while (d > 0 && isdigit(static_cast<unsigned char>(f[d]))) d--;
  return f[d] == ':';
}

int Spool(FILE *in, const char *what, char *path, int cap) {
  const char *dir = getenv("TMPDIR");
  std::string tmpl = std::string(dir && *dir ? dir : "/tmp") + "/kamd_rx_XXXXXX";
  std::vector<char> name(tmpl.begin(), tmpl.end());
  name.push_back('\0');
  const int fd = mkstemp(name.data());
  if (fd < 0) return kamd::SetError(KAMD_ERR_ARG, "cannot create a temporary file for %s", what);
  FILE *out = fdopen(fd, "wb");
  char buf[1 << 16];
  size_t n;
  bool ok = out != NULL;
  while (ok && (n = fread(buf, 1, sizeof(buf), in)) > 0) ok = fwrite(buf, 1, n, out) == n;
  if (out) ok = fclose(out) == 0 && ok;
  if (!ok) { unlink(name.data()); return kamd::SetError(KAMD_ERR_ARG, "spooling %s failed", what); }
  if (CopyOut(name.data(), path, cap)) { unlink(name.data()); return KAMD_ERR_ARG; }
  return KAMD_OK;
}

}  // namespace

extern "C" {

int kamd_classify_rxfilename(const char *filename) {
  const std::string f(filename ? filename : "");
  if (f.empty() || f == "-") return KAMD_RX_STDIN;
  const unsigned char first = f[0], last = f.back();
  if (first == '|') return KAMD_RX_NONE;
  if (last == '|') return KAMD_RX_PIPE;
  if (isspace(first) || isspace(last)) return KAMD_RX_NONE;
  if (LooksLikeSpecifier(f)) return KAMD_RX_NONE;
  if (EndsInOffset(f)) return KAMD_RX_OFFSET_FILE;
  if (f.find('|') != std::string::npos) return KAMD_RX_NONE;
  return KAMD_RX_FILE;
}

int kamd_classify_wxfilename(const char *filename) {
  const std::string f(filename ? filename : "");
  if (f.empty() || f == "-") return KAMD_WX_STDOUT;
  const unsigned char first = f[0], last = f.back();
  if (first == '|') return KAMD_WX_PIPE;
  if (isspace(first) || isspace(last) || last == '|') return KAMD_WX_NONE;
  if (LooksLikeSpecifier(f)) return KAMD_WX_NONE;
  if (EndsInOffset(f)) return KAMD_WX_NONE;
  if (f.find('|') != std::string::npos) return KAMD_WX_NONE;
  return KAMD_WX_FILE;
}

int kamd_classify_rspecifier(const char *rspecifier, char *rxfilename, int cap, int *opts) {
  std::string rx;
  int o = 0;
  const int t = RspecType(rspecifier ? rspecifier : "", &rx, &o);
  if (opts) *opts = o;
  if (CopyOut(rx, rxfilename, cap)) return -1;
  return t;
}

int kamd_classify_wspecifier(const char *wspecifier, char *archive_wxfilename, int ark_cap, char *script_wxfilename,
                             int scp_cap, int *opts) {
  std::string ark, scp;
  int o = KAMD_WSPEC_BINARY;
  const int t = WspecType(wspecifier ? wspecifier : "", &ark, &scp, &o);
  if (opts) *opts = o;
  if (CopyOut(ark, archive_wxfilename, ark_cap) || CopyOut(scp, script_wxfilename, scp_cap)) return -1;
  return t;
}

int kamd_rx_materialize(const char *rxfilename, char *path, int cap, int64_t *offset, int *is_temp) {
  const std::string f(rxfilename ? rxfilename : "");
  *offset = 0; *is_temp = 0;
  switch (kamd_classify_rxfilename(f.c_str())) {
    case KAMD_RX_FILE:
      return CopyOut(f, path, cap) ? KAMD_ERR_ARG : KAMD_OK;
    case KAMD_RX_OFFSET_FILE: {
      const size_t colon = f.rfind(':');
      *offset = strtoll(f.c_str() + colon + 1, NULL, 10);
      return CopyOut(f.substr(0, colon), path, cap) ? KAMD_ERR_ARG : KAMD_OK;
    }
    case KAMD_RX_STDIN: {
      const int rc = Spool(stdin, "standard input", path, cap);
      if (rc == KAMD_OK) *is_temp = 1;
      return rc;
    }
    case KAMD_RX_PIPE: {
      const std::string cmd = f.substr(0, f.size() - 1);            // PipeInputImpl::Open (kaldi-io.cc:470-500)
      FILE *p = popen(cmd.c_str(), "r");
      if (!p) return kamd::SetError(KAMD_ERR_ARG, "Failed opening pipe for reading, command is: %s", cmd.c_str());
      const int rc = Spool(p, cmd.c_str(), path, cap);
      const int status = pclose(p);
      if (rc != KAMD_OK) return rc;
      *is_temp = 1;
      if (status != 0) { unlink(path); *is_temp = 0; return kamd::SetError(KAMD_ERR_ARG, "Pipe %s had nonzero return status %d", cmd.c_str(), status); }
      return KAMD_OK;
    }
    default:
      return kamd::SetError(KAMD_ERR_ARG, "Invalid input filename format %s", f.c_str());
  }
}

}  // extern "C"
