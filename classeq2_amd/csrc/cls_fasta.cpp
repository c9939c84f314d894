// FASTA input stage of the placement path, with the reference's exact record
// semantics: FileOrStdin::sequence_content_by_channel
// (core/src/domain/dtos/file_or_stdin.rs:76-116) and
// SequenceBody::remove_non_iupac_from_sequence (core/src/domain/dtos/sequence.rs:47-56).
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "cls_place.h"

namespace {

// std::io::BufRead::lines() yields Err on a line that is not valid UTF-8; the
// reference propagates it (`line?`) and the caller drops it (mod.rs:119).
bool valid_utf8(const unsigned char* s, size_t n) {
    size_t i = 0;
    while (i < n) {
        unsigned char c = s[i];
        if (c < 0x80) { ++i; continue; }
        size_t need;
        uint32_t cp;
        if ((c & 0xE0) == 0xC0) { need = 1; cp = c & 0x1F; if (cp < 2) return false; }
        else if ((c & 0xF0) == 0xE0) { need = 2; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { need = 3; cp = c & 0x07; if (cp > 4) return false; }
        else return false;
        if (i + need >= n) return false;  // truncated multi-byte sequence
        for (size_t k = 1; k <= need; ++k) {
            unsigned char d = s[i + k];
            if ((d & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (d & 0x3F);
        }
        if (need == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
        if (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
        i += need + 1;
    }
    return true;
}

}  // namespace

extern "C" void cls_fasta_free(cls_fasta* f) {
    if (!f) return;
    free(f->headers); free(f->header_off); free(f->bases); free(f->base_off);
    memset(f, 0, sizeof *f);
}

extern "C" int cls_fasta_parse(const char* text, size_t len, cls_fasta* out) {
    if (!out || (!text && len)) return CLS_E_INVALID_ARG;
    memset(out, 0, sizeof *out);
    try {
        std::string headers, bases, header, sequence;
        std::vector<uint64_t> hoff{0}, boff{0};
        bool truncated = false;
        auto emit = [&]() {
            headers += header; hoff.push_back(headers.size());
            bases += sequence; boff.push_back(bases.size());
        };
        size_t pos = 0;
        while (pos < len) {
            const char* nl = (const char*)memchr(text + pos, '\n', len - pos);
            size_t end = nl ? (size_t)(nl - text) : len;
            size_t lend = end;
            if (nl && lend > pos && text[lend - 1] == '\r') --lend;  // lines() strips "\n" or "\r\n"
            const char* line = text + pos;
            size_t ll = lend - pos;
            pos = nl ? end + 1 : len;
            if (!valid_utf8((const unsigned char*)line, ll)) { truncated = true; break; }
            if (ll == 0) continue;                                    // :87-89
            if (line[0] == '>') {
                if (!header.empty()) {                                // :92-95 (emitted even if the sequence is empty)
                    emit();
                    sequence.clear();
                } else if (!sequence.empty()) {                       // :96-100
                    truncated = true;
                    break;
                }
                header.clear();
                for (size_t i = 0; i < ll; ++i) if (line[i] != '>') header.push_back(line[i]);  // replace(">", "") :102
            } else {
                for (size_t i = 0; i < ll; ++i) {                     // sequence.rs:47-56
                    char c = line[i];
                    if (c >= 'a' && c <= 'z') c = (char)(c - 32);
                    if (c == 'A' || c == 'C' || c == 'G' || c == 'T') sequence.push_back(c);
                }
            }
        }
        if (!truncated && !header.empty() && !sequence.empty()) emit();  // :111-113
        out->n = (uint32_t)(hoff.size() - 1);
        out->truncated = truncated ? 1 : 0;
        out->headers = (char*)malloc(headers.size() + 1);
        out->bases = (char*)malloc(bases.size() + 1);
        out->header_off = (uint64_t*)malloc(hoff.size() * 8);
        out->base_off = (uint64_t*)malloc(boff.size() * 8);
        if (!out->headers || !out->bases || !out->header_off || !out->base_off) { cls_fasta_free(out); return CLS_E_NOMEM; }
        memcpy(out->headers, headers.data(), headers.size());
        memcpy(out->bases, bases.data(), bases.size());
        memcpy(out->header_off, hoff.data(), hoff.size() * 8);
        memcpy(out->base_off, boff.data(), boff.size() * 8);
        return CLS_OK;
    } catch (const std::bad_alloc&) {
        cls_fasta_free(out);
        return CLS_E_NOMEM;
    } catch (...) {
        cls_fasta_free(out);
        return CLS_E_INTERNAL;
    }
}
