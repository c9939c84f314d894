// Host-side mirror of the reference's batch driver around the placement path:
//   core::use_cases::place_sequences          (core/src/use_cases/place_sequences/mod.rs:43-270)
//   PlacementResponse / PlacementStatus serde  (core/src/domain/dtos/placement_response.rs:30-94)
//   Clade / Annotation shapes                  (clade.rs:18-38, annotation.rs:5-34)
// in C++ (the reference is compiled Rust; no Rust toolchain in this image).  The tree/index come from the
// reference's JSON export (`cls convert database -f json`, ports/cli/src/cmds/convert.rs:161-205); the
// placements come from the HIP kernels through cls_place_batch.  Nothing here touches the oracle.
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "cls_host.h"
#include "cls_host_internal.h"
#include "cls_json.h"
#include "cls_murmur.h"
#include "cls_tuning.h"

namespace cls_host {

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }

// ---- f64 formatting exactly as Rust's ryu (serde_json / serde_yaml floats) ------------------------
std::string fmt_f64(double v) {
    if (v != v) return ".nan";
    if (v == 1.0 / 0.0) return ".inf";
    if (v == -1.0 / 0.0) return "-.inf";
    if (v == 0.0) return (1.0 / v < 0) ? "-0.0" : "0.0";
    char buf[64];
    auto res = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);  // shortest round-trip digits
    std::string sci(buf, res.ptr);
    std::string out;
    size_t i = 0;
    if (sci[0] == '-') { out.push_back('-'); i = 1; }
    std::string digits;
    size_t epos = sci.find('e');
    for (size_t j = i; j < epos; ++j) if (sci[j] != '.') digits.push_back(sci[j]);
    int exp10 = atoi(sci.c_str() + epos + 1);  // value = d.ddd * 10^exp10
    const int k = (int)digits.size();
    const int kk = exp10 + 1;  // position of the decimal point relative to the digit string
    if (k <= kk && kk <= 16) {  // 1234e7 -> 12340000000.0
        out += digits;
        out.append((size_t)(kk - k), '0');
        out += ".0";
    } else if (0 < kk && kk <= 16) {  // 1234e-2 -> 12.34
        out += digits.substr(0, (size_t)kk);
        out.push_back('.');
        out += digits.substr((size_t)kk);
    } else if (-5 < kk && kk <= 0) {  // 1234e-6 -> 0.001234
        out += "0.";
        out.append((size_t)(-kk), '0');
        out += digits;
    } else if (k == 1) {  // 1e30
        out += digits;
        out.push_back('e');
        out += std::to_string(kk - 1);
    } else {  // 1234e30 -> 1.234e33
        out.push_back(digits[0]);
        out.push_back('.');
        out += digits.substr(1);
        out.push_back('e');
        out += std::to_string(kk - 1);
    }
    return out;
}

// (data model: cls_host_internal.h)
const char* kind_name(int k) { return k == CLS_KIND_ROOT ? "ROOT" : k == CLS_KIND_LEAF ? "LEAF" : "NODE"; }

Clade clade_from_json(const cls::JVal& j) {
    if (j.kind != cls::JVal::Obj) throw std::runtime_error("clade: expected an object");
    Clade c;
    const cls::JVal* v;
    if (!(v = j.get("id"))) throw std::runtime_error("clade without id");
    c.id = v->as_u64();
    if ((v = j.get("parent")) && !v->is_null()) { c.has_parent = true; c.parent = v->as_u64(); }
    if (!(v = j.get("kind")) || v->kind != cls::JVal::Str) throw std::runtime_error("clade without kind");
    c.kind = v->s == "ROOT" ? CLS_KIND_ROOT : v->s == "LEAF" ? CLS_KIND_LEAF : v->s == "NODE" ? CLS_KIND_NODE : throw std::runtime_error("bad clade kind " + v->s);
    if ((v = j.get("name")) && !v->is_null()) { c.has_name = true; c.name = v->s; }
    if ((v = j.get("support")) && !v->is_null()) { c.has_support = true; c.support = v->as_f64(); }
    if ((v = j.get("length")) && !v->is_null()) { c.has_length = true; c.length = v->as_f64(); }
    if ((v = j.get("children")) && !v->is_null()) {
        c.has_children = true;
        for (auto& ch : v->arr) c.children.push_back(clade_from_json(ch));
    }
    return c;
}

void flatten(cls_tree* t) {
    t->rows.clear(); t->row_clade.clear(); t->first_row_of_id.clear();
    std::vector<const Clade*> q{&t->root};
    for (size_t i = 0; i < q.size(); ++i) {
        const Clade* c = q[i];
        cls_node n;
        memset(&n, 0, sizeof n);
        n.id = c->id;
        n.parent = c->has_parent ? c->parent : CLS_NO_PARENT;
        n.first_child = c->children.empty() ? 0 : (uint32_t)q.size();
        n.n_children = (uint32_t)c->children.size();
        n.kind = (uint8_t)c->kind;
        n.has_children = c->has_children ? 1 : 0;
        t->rows.push_back(n);
        t->row_clade.push_back(c);
        for (auto& ch : c->children) q.push_back(&ch);
    }
    // DFS order for first-match semantics
    std::vector<const Clade*> st{&t->root};
    std::map<const Clade*, uint32_t> row_of;
    for (uint32_t r = 0; r < t->row_clade.size(); ++r) row_of[t->row_clade[r]] = r;
    while (!st.empty()) {
        const Clade* c = st.back(); st.pop_back();
        t->first_row_of_id.emplace(c->id, row_of[c]);  // emplace keeps the first
        for (auto it = c->children.rbegin(); it != c->children.rend(); ++it) st.push_back(&*it);
    }
}

// ---- annotations: the YAML subset of tests/models/bsub-gyrb-annotations.yaml ---------------------
// - clade: N / meta: / - !Tag scalar | block scalar.  (serde_yaml of Vec<Annotation>, annotation.rs:25-34)
std::string rstrip(const std::string& s) { size_t e = s.find_last_not_of(" \t\r"); return e == std::string::npos ? "" : s.substr(0, e + 1); }

void parse_annotations_yaml(const std::string& text, std::vector<Annotation>& out) {
    std::vector<std::string> lines;
    { std::stringstream ss(text); std::string l; while (std::getline(ss, l)) lines.push_back(l); }
    auto indent_of = [](const std::string& l) { size_t i = 0; while (i < l.size() && l[i] == ' ') ++i; return i; };
    for (size_t i = 0; i < lines.size();) {
        std::string l = rstrip(lines[i]);
        if (l.empty() || l[indent_of(l)] == '#') { ++i; continue; }
        size_t ind = indent_of(l);
        std::string body = l.substr(ind);
        if (body.rfind("- clade:", 0) == 0) {
            Annotation a;
            a.clade = (uint32_t)strtoul(body.c_str() + 8, nullptr, 10);
            out.push_back(a);
            ++i;
        } else if (body.rfind("clade:", 0) == 0 && !out.empty()) { out.back().clade = (uint32_t)strtoul(body.c_str() + 6, nullptr, 10); ++i; }
        else if (body == "meta:" || body == "- meta:") {
            if (body[0] == '-') out.emplace_back();
            if (out.empty()) throw std::runtime_error("annotations: meta before any clade");
            out.back().has_meta = true;
            ++i;
        } else if (body.rfind("- !", 0) == 0) {
            if (out.empty()) throw std::runtime_error("annotations: tag before any clade");
            size_t sp = body.find(' ', 3);
            Tag t;
            t.name = body.substr(3, sp == std::string::npos ? std::string::npos : sp - 3);
            std::string val = sp == std::string::npos ? "" : body.substr(sp + 1);
            ++i;
            if (val == "|" || val == "|-" || val == "|+" || val == ">") {
                // block scalar: lines more indented than the item; literal style keeps line breaks
                std::string acc;
                size_t bind = std::string::npos;
                size_t pending_blank = 0;
                bool any = false;
                while (i < lines.size()) {
                    std::string bl = lines[i];
                    if (rstrip(bl).empty()) { ++pending_blank; ++i; continue; }
                    size_t bi = indent_of(bl);
                    if (bi <= ind) break;
                    if (bind == std::string::npos) bind = bi;
                    if (any) acc.push_back('\n');
                    acc.append(pending_blank, '\n');
                    pending_blank = 0;
                    acc += rstrip(bl).substr(std::min(bind, bl.size()));
                    any = true;
                    ++i;
                }
                if (val != "|-") acc.push_back('\n');  // clip (|) keeps one trailing newline
                t.sval = acc;
            } else {
                if (val.size() >= 2 && ((val.front() == '\'' && val.back() == '\'') || (val.front() == '"' && val.back() == '"'))) {
                    const char q = val.front();
                    std::string in = val.substr(1, val.size() - 2), o;
                    for (size_t z = 0; z < in.size(); ++z) {
                        if (q == '\'' && in[z] == '\'' && z + 1 < in.size() && in[z + 1] == '\'') { o.push_back('\''); ++z; }
                        else if (q == '"' && in[z] == '\\' && z + 1 < in.size()) { ++z; o.push_back(in[z] == 'n' ? '\n' : in[z] == 't' ? '\t' : in[z]); }
                        else o.push_back(in[z]);
                    }
                    val = o;
                }
                t.sval = val;
            }
            if (t.name == "Taxid") { t.is_int = true; t.ival = strtoull(t.sval.c_str(), nullptr, 10); }
            out.back().meta.push_back(t);
        } else throw std::runtime_error("annotations: unsupported YAML line: " + l);
    }
}

// ---- YAML / JSON emission ---------------------------------------------------------------------------
bool yaml_ambiguous(const std::string& s) {
    static const char* kw[] = {"", "~", "null", "Null", "NULL", "true", "True", "TRUE", "false", "False", "FALSE", "y", "Y", "yes", "Yes", "YES",
                               "n", "N", "no", "No", "NO", "on", "On", "ON", "off", "Off", "OFF", ".nan", ".NaN", ".NAN", ".inf", ".Inf", ".INF",
                               "-.inf", "-.Inf", "-.INF", "+.inf", "+.Inf", "+.INF"};
    for (auto k : kw) if (s == k) return true;
    // number-like?
    char* end = nullptr;
    errno = 0;
    strtod(s.c_str(), &end);
    if (end && *end == 0 && !s.empty() && (isdigit((unsigned char)s[0]) || s[0] == '-' || s[0] == '+' || s[0] == '.')) return true;
    if (s.size() > 2 && s[0] == '0' && (s[1] == 'x' || s[1] == 'o')) return true;
    return false;
}

bool yaml_needs_quotes(const std::string& s) {
    if (yaml_ambiguous(s)) return true;
    const char c0 = s[0];
    if (strchr("-?:,[]{}#&*!|>'\"%@`", c0)) {
        if ((c0 == '-' || c0 == '?' || c0 == ':') && s.size() > 1 && s[1] != ' ') { /* allowed as plain */ }
        else return true;
    }
    if (s.front() == ' ' || s.back() == ' ') return true;
    for (size_t i = 0; i < s.size(); ++i) {
        const unsigned char c = (unsigned char)s[i];
        if (c == ':' && (i + 1 == s.size() || s[i + 1] == ' ')) return true;
        if (c == '#' && i > 0 && s[i - 1] == ' ') return true;
        if (c < 0x20 || c == 0x7F) return true;
    }
    return false;
}

bool yaml_needs_double(const std::string& s) {
    for (unsigned char c : s) if ((c < 0x20 && c != '\n') || c == 0x7F) return true;
    return false;
}

// scalar after "key: " or "- " at `indent` (the indentation of that key / item)
void yaml_str(std::string& o, const std::string& s, size_t indent) {
    if (s.find('\n') != std::string::npos && !yaml_needs_double(s)) {  // literal block, like serde_yaml does for multi-line strings
        size_t trail = 0;
        while (trail < s.size() && s[s.size() - 1 - trail] == '\n') ++trail;
        o += trail == 1 ? "|" : trail == 0 ? "|-" : "|+";
        o.push_back('\n');
        size_t pos = 0;
        const std::string body = s.substr(0, s.size() - (trail ? 1 : 0));
        while (pos <= body.size()) {
            size_t nl = body.find('\n', pos);
            std::string line = body.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
            if (!line.empty()) { o.append(indent + 2, ' '); o += line; }
            o.push_back('\n');
            if (nl == std::string::npos) break;
            pos = nl + 1;
        }
        return;
    }
    if (yaml_needs_double(s)) {
        o.push_back('"');
        for (unsigned char c : s) {
            if (c == '"') o += "\\\""; else if (c == '\\') o += "\\\\"; else if (c == '\n') o += "\\n"; else if (c == '\t') o += "\\t";
            else if (c < 0x20 || c == 0x7F) { char b[8]; snprintf(b, sizeof b, "\\x%02X", c); o += b; } else o.push_back((char)c);
        }
        o += "\"\n";
    } else if (yaml_needs_quotes(s)) {
        o.push_back('\'');
        for (char c : s) { if (c == '\'') o += "''"; else o.push_back(c); }
        o += "'\n";
    } else {
        o += s;
        o.push_back('\n');
    }
}

void yaml_clade(std::string& o, const Clade& c, size_t ind, bool first_inline) {
    // first_inline: the first key follows "- " on the same line (sequence item)
    auto key = [&](const char* k, bool first) { if (!(first && first_inline)) o.append(ind, ' '); o += k; };
    key("id: ", true); o += std::to_string(c.id); o.push_back('\n');
    key("parent: ", false); o += c.has_parent ? std::to_string(c.parent) : "null"; o.push_back('\n');
    key("kind: ", false); o += kind_name(c.kind); o.push_back('\n');
    if (c.has_name) { key("name: ", false); yaml_str(o, c.name, ind); }
    if (c.has_support) { key("support: ", false); o += fmt_f64(c.support); o.push_back('\n'); }
    if (c.has_length) { key("length: ", false); o += fmt_f64(c.length); o.push_back('\n'); }
    if (c.has_children) {
        if (c.children.empty()) { key("children: []\n", false); }
        else {
            key("children:\n", false);
            for (auto& ch : c.children) { o.append(ind, ' '); o += "- "; yaml_clade(o, ch, ind + 2, true); }
        }
    }
}

void json_str(std::string& o, const std::string& s) {
    o.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
                else o.push_back((char)c);
        }
    }
    o.push_back('"');
}

std::string json_f64(double v) {
    if (v != v || v == 1.0 / 0.0 || v == -1.0 / 0.0) return "null";  // serde_json writes non-finite floats as null
    return fmt_f64(v);
}

void json_clade(std::string& o, const Clade& c) {
    o += "{\"id\":" + std::to_string(c.id) + ",\"parent\":" + (c.has_parent ? std::to_string(c.parent) : "null") + ",\"kind\":\"" + kind_name(c.kind) + "\"";
    if (c.has_name) { o += ",\"name\":"; json_str(o, c.name); }
    if (c.has_support) o += ",\"support\":" + json_f64(c.support);
    if (c.has_length) o += ",\"length\":" + json_f64(c.length);
    if (c.has_children) {
        o += ",\"children\":[";
        for (size_t i = 0; i < c.children.size(); ++i) { if (i) o.push_back(','); json_clade(o, c.children[i]); }
        o.push_back(']');
    }
    o.push_back('}');
}

std::string rust_debug_str(const std::string& s) {  // `{:?}` of SequenceHeader(String) (place_sequence.rs:131-134)
    std::string o = "SequenceHeader(\"";
    for (unsigned char c : s) {
        if (c == '"') o += "\\\""; else if (c == '\\') o += "\\\\"; else if (c == '\n') o += "\\n"; else if (c == '\r') o += "\\r";
        else if (c == '\t') o += "\\t"; else if (c == 0) o += "\\0";
        else if (c < 0x20 || c == 0x7F) { char b[16]; snprintf(b, sizeof b, "\\u{%x}", c); o += b; } else o.push_back((char)c);
    }
    return o + "\")";
}

// PlacementStatus::to_string (placement_response.rs:30-42) from a record
std::string code_of(const cls_placement& r, const std::string& header) {
    switch (r.status) {
        case CLS_UNCLASSIFIABLE_NO_MATCH: return "Unclassifiable: Query sequence " + rust_debug_str(header) + " may not be related to the phylogeny";
        case CLS_UNCLASSIFIABLE_NO_ROOT: return "Unclassifiable: Query sequence has no overlapping kmers with the reference tree";
        case CLS_UNCLASSIFIABLE_COVERAGE: return "Unclassifiable: Insufficient kmers coverage: " + std::to_string(r.one);
        case CLS_UNCLASSIFIABLE_LEVEL1: return "Unclassifiable: Tree introspection not possible. Query sequence has no overlapping kmers with the reference tree";
        case CLS_IDENTITY_FOUND: return "IdentityFound";
        case CLS_MAX_RESOLUTION: return "MaxResolutionReached: LCA Accepted";
        case CLS_INCONCLUSIVE: return "Inconclusive: Multiple proposals";
        default: return "";
    }
}

// error text for <out>.error (MappedErrors Display of mycelium-base is not under /root/reference: "parity unpinned")
const char* error_text(uint8_t status) {
    switch (status) {
        case CLS_ERR_TOO_FEW_KMERS: return "The sequence does not contain enough kmers.";
        case CLS_ERR_MAX_ITER: return "The maximum number of iterations has been reached.";
        case CLS_ERR_ROOT_NO_CHILDREN: return "The root node does not have children. This is unexpected.";
        case CLS_ERR_INVALID_BASE: return "Invalid character in sequence";
        case CLS_ERR_READ_TOO_LONG: return "The sequence exceeds the k-mer capacity of the GPU engine.";
        default: return nullptr;
    }
}

// annotations whose clade lies on the path placed node -> root, sorted by clade (mod.rs:187-224)
std::vector<const Annotation*> annotations_for(const cls_tree* t, uint64_t clade_id) {
    std::vector<const Annotation*> res;
    auto it = t->first_row_of_id.find(clade_id);
    if (it == t->first_row_of_id.end()) return res;
    std::set<uint64_t> path;  // Clade::get_path_to_root (clade.rs:111-125): follows the `parent` FIELDS via get_node_by_id
    const Clade* c = t->row_clade[it->second];
    for (int guard = 0; c && guard < 1000000; ++guard) {
        path.insert(c->id);
        if (!c->has_parent) break;
        path.insert(c->parent);
        auto p = t->first_row_of_id.find(c->parent);
        c = p == t->first_row_of_id.end() ? nullptr : t->row_clade[p->second];
        if (c && path.count(c->id) && c->has_parent && path.count(c->parent)) break;  // parent cycle in a malformed file
    }
    for (auto& a : t->annotations) if (path.count((uint64_t)a.clade)) res.push_back(&a);
    std::stable_sort(res.begin(), res.end(), [](const Annotation* a, const Annotation* b) { return a->clade < b->clade; });
    return res;
}

void serialize_one(const cls_tree* t, const std::string& header, const cls_placement& r, int format, std::string& o) {
    const std::string code = code_of(r, header);
    const bool has_placement = r.status == CLS_IDENTITY_FOUND || r.status == CLS_MAX_RESOLUTION || r.status == CLS_INCONCLUSIVE;
    std::vector<const Annotation*> ann;
    bool ann_some = false;
    if (t->has_annotations && (r.status == CLS_IDENTITY_FOUND || r.status == CLS_MAX_RESOLUTION)) {  // clade_from_placement_status.rs:5-19
        ann = annotations_for(t, r.clade_id);
        ann_some = !ann.empty();
    }
    const Clade* placed = nullptr;
    if (r.status == CLS_IDENTITY_FOUND) {
        auto it = t->first_row_of_id.find(r.clade_id);
        if (it != t->first_row_of_id.end()) placed = t->row_clade[it->second];
    }
    if (format == CLS_FORMAT_YAML) {
        o += "---\n";
        o += "query: "; yaml_str(o, header, 0);
        o += "code: "; yaml_str(o, code, 0);
        if (ann_some) {
            o += "annotations:\n";
            for (auto* a : ann) {
                o += "- clade: " + std::to_string(a->clade) + "\n";
                if (a->has_meta) {
                    if (a->meta.empty()) o += "  meta: []\n";
                    else {
                        o += "  meta:\n";
                        for (auto& tg : a->meta) {
                            o += "  - !" + tg.name + " ";
                            if (tg.is_int) o += std::to_string(tg.ival) + "\n"; else yaml_str(o, tg.sval, 2);
                        }
                    }
                }
            }
        }
        if (has_placement) {
            if (r.status == CLS_MAX_RESOLUTION) o += "placement: " + std::to_string(r.clade_id) + "\n";
            else if (r.status == CLS_IDENTITY_FOUND) {
                o += "placement:\n  clade:\n";
                if (placed) yaml_clade(o, *placed, 4, false);
                o += "  one: " + std::to_string(r.one) + "\n  rest: " + std::to_string(r.rest) + "\n";
            } else yaml_str((o += "placement: "), code, 0);
        }
    } else {
        o += "{\"query\":"; json_str(o, header);
        o += ",\"code\":"; json_str(o, code);
        if (ann_some) {
            o += ",\"annotations\":[";
            for (size_t i = 0; i < ann.size(); ++i) {
                if (i) o.push_back(',');
                o += "{\"clade\":" + std::to_string(ann[i]->clade);
                if (ann[i]->has_meta) {
                    o += ",\"meta\":[";
                    for (size_t j = 0; j < ann[i]->meta.size(); ++j) {
                        if (j) o.push_back(',');
                        const Tag& tg = ann[i]->meta[j];
                        o += "{"; json_str(o, tg.name); o.push_back(':');
                        if (tg.is_int) o += std::to_string(tg.ival); else json_str(o, tg.sval);
                        o += "}";
                    }
                    o += "]";
                }
                o += "}";
            }
            o += "]";
        }
        if (has_placement) {
            o += ",\"placement\":";
            if (r.status == CLS_MAX_RESOLUTION) o += std::to_string(r.clade_id);
            else if (r.status == CLS_IDENTITY_FOUND) {
                o += "{\"clade\":";
                if (placed) json_clade(o, *placed); else o += "null";
                o += ",\"one\":" + std::to_string(r.one) + ",\"rest\":" + std::to_string(r.rest) + "}";
            } else json_str(o, code);
        }
        o += "}\n";
    }
}

std::string read_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path + ": " + strerror(errno));
    std::string s;
    struct stat sb;
    if (fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {  // regular file: one read into place
        s.resize((size_t)sb.st_size);
        const size_t got = fread(&s[0], 1, s.size(), f);
        s.resize(got);
    }
    char buf[1 << 16];  // pipes, files that grew, ...
    for (size_t got; (got = fread(buf, 1, sizeof buf, f)) > 0;) s.append(buf, got);
    fclose(f);
    return s;
}

std::string with_extension(const std::string& path, const char* ext) {  // PathBuf::set_extension
    size_t slash = path.find_last_of('/');
    size_t dot = path.find_last_of('.');
    std::string stem = (dot != std::string::npos && (slash == std::string::npos || dot > slash + 1)) ? path.substr(0, dot) : path;
    return stem + "." + ext;
}

}  // namespace cls_host

using namespace cls_host;

extern "C" const char* cls_host_last_error(void) { return g_err.c_str(); }

int cls_host_fail(int code, const std::string& msg) { return fail(code, msg); }

void cls_tree_visit_leaves(const cls_tree* t, const std::function<void(const char*, const std::vector<uint64_t>&)>& fn) {
    struct Fr { const Clade* c; size_t next; };
    std::vector<Fr> st{{&t->root, 0}};
    std::vector<uint64_t> path{t->root.id};
    if (t->root.kind == CLS_KIND_LEAF) fn(t->root.has_name ? t->root.name.c_str() : nullptr, path);
    while (!st.empty()) {
        Fr& f = st.back();
        if (f.c->kind == CLS_KIND_LEAF || f.next >= f.c->children.size()) { st.pop_back(); path.pop_back(); continue; }  // leaves are not descended into
        const Clade* ch = &f.c->children[f.next++];
        path.push_back(ch->id);
        if (ch->kind == CLS_KIND_LEAF) fn(ch->has_name ? ch->name.c_str() : nullptr, path);
        st.push_back({ch, 0});
    }
}

void cls_tree_set_kmers_map(cls_tree* t, uint64_t k, uint64_t m, std::vector<uint64_t>&& bucket_key,
                            std::vector<uint64_t>&& bucket_kmer_off, std::vector<uint64_t>&& kmer_hash,
                            std::vector<uint64_t>&& kmer_node_off, std::vector<uint64_t>&& node_ids) {
    t->has_kmers = true;
    t->k_size = k;
    t->m_size = m;
    t->bucket_key = std::move(bucket_key);
    t->bucket_kmer_off = std::move(bucket_kmer_off);
    if (t->bucket_kmer_off.empty()) t->bucket_kmer_off.push_back(0);
    t->kmer_hash = std::move(kmer_hash);
    t->kmer_node_off = std::move(kmer_node_off);
    t->node_ids = std::move(node_ids);
}

extern "C" void cls_tree_free(cls_tree* t) { delete t; }

namespace cls_host {
void tree_from_doc(const cls::JVal& doc, cls_tree* t) {
    const cls::JVal* root = doc.get("root");
    t->root = clade_from_json(root ? *root : doc);  // `--only-tree` exports hold the root clade alone
    flatten(t);
    const cls::JVal* v;
    if (root) {
        t->has_header = true;
        if ((v = doc.get("id")) && !v->is_null()) t->uuid = v->s;
        if ((v = doc.get("name")) && !v->is_null()) t->name = v->s;
        if ((v = doc.get("minBranchSupport")) && !v->is_null()) t->min_branch_support = v->kind == cls::JVal::Num ? v->as_f64() : strtod(v->s.c_str(), nullptr);
        if ((v = doc.get("inMemorySize")) && !v->is_null()) { t->has_in_memory_size = true; t->in_memory_size = v->s; }
    }
    const cls::JVal* km = root ? doc.get("kmersMap") : nullptr;
    if (km && !km->is_null()) {
        t->has_kmers = true;
        t->k_size = km->get("kSize") ? km->get("kSize")->as_u64() : 0;
        t->m_size = km->get("mSize") ? km->get("mSize")->as_u64() : 0;
        const cls::JVal* map = km->get("map");
        t->bucket_kmer_off.assign(1, 0);
        t->kmer_node_off.assign(1, 0);
        if (map && !map->obj.empty() && map->obj.front().second.kind == cls::JVal::Arr) {
            // files written before the minimizer buckets existed (core/src/tests/data/.../outputs/*.yaml):
            // map = {k-mer string: [node ids]}; one bucket, keyed like mSize = 0 (kmers_map.rs:131-134)
            t->m_size = 0;
            t->bucket_key.push_back(0);
            for (auto& kv : map->obj) {
                t->kmer_hash.push_back(cls::murmur3_h1_bytes(kv.first.data(), (uint32_t)kv.first.size()));
                for (auto& id : kv.second.arr) t->node_ids.push_back(id.as_u64());
                t->kmer_node_off.push_back(t->node_ids.size());
            }
            t->bucket_kmer_off.push_back(t->kmer_hash.size());
        } else if (map) for (auto& b : map->obj) {
            t->bucket_key.push_back(strtoull(b.first.c_str(), nullptr, 10));
            for (auto& kv : b.second.obj) {
                t->kmer_hash.push_back(strtoull(kv.first.c_str(), nullptr, 10));
                for (auto& id : kv.second.arr) t->node_ids.push_back(id.as_u64());
                t->kmer_node_off.push_back(t->node_ids.size());
            }
            t->bucket_kmer_off.push_back(t->kmer_hash.size());
        }
    }
    const cls::JVal* an = root ? doc.get("annotations") : nullptr;
    if (an && an->kind == cls::JVal::Arr && !an->arr.empty()) {
        t->has_annotations = true;
        for (auto& a : an->arr) {
            Annotation x;
            x.clade = (uint32_t)a.get("clade")->as_u64();
            if (const cls::JVal* m = a.get("meta")) if (!m->is_null()) {
                x.has_meta = true;
                for (auto& tg : m->arr) for (auto& kv : tg.obj) {
                    Tag g; g.name = kv.first;
                    if (kv.second.kind == cls::JVal::Num) { g.is_int = true; g.ival = kv.second.as_u64(); } else g.sval = kv.second.s;
                    x.meta.push_back(g);
                }
            }
            t->annotations.push_back(x);
        }
    }
}
}  // namespace cls_host

extern "C" int cls_tree_load_json(const char* path, cls_tree** out) {
    if (!path || !out) return fail(CLS_E_INVALID_ARG, "cls_tree_load_json: null argument");
    try {
        std::string text = read_file(path);
        cls::JVal doc = cls::JParser(text.data(), text.size()).parse();
        auto t = std::make_unique<cls_tree>();
        tree_from_doc(doc, t.get());
        *out = t.release();
        return CLS_OK;
    } catch (const std::exception& e) {
        return fail(CLS_E_BAD_DB, std::string("cls_tree_load_json: ") + e.what());
    } catch (...) {
        return fail(CLS_E_INTERNAL, "cls_tree_load_json: unknown exception");
    }
}

extern "C" int cls_tree_set_annotations_yaml(cls_tree* t, const char* path) {
    if (!t || !path) return fail(CLS_E_INVALID_ARG, "cls_tree_set_annotations_yaml: null argument");
    try {
        std::vector<Annotation> a;
        parse_annotations_yaml(read_file(path), a);
        if (!a.empty()) { t->annotations = std::move(a); t->has_annotations = true; }  // place_sequences.rs:137-144
        return CLS_OK;
    } catch (const std::exception& e) {
        return fail(CLS_E_BAD_DB, std::string("cls_tree_set_annotations_yaml: ") + e.what());
    } catch (...) {
        return fail(CLS_E_INTERNAL, "cls_tree_set_annotations_yaml: unknown exception");
    }
}

extern "C" int cls_tree_desc(const cls_tree* t, cls_db_desc* d) {
    if (!t || !d) return fail(CLS_E_INVALID_ARG, "cls_tree_desc: null argument");
    if (!t->has_kmers) return fail(CLS_E_BAD_DB, "cls_tree_desc: the file holds no kmersMap (tree-only export)");
    memset(d, 0, sizeof *d);
    d->abi_version = CLS_ABI_VERSION;
    d->n_nodes = (uint32_t)t->rows.size();
    d->nodes = t->rows.data();
    d->k_size = t->k_size;
    d->m_size = t->m_size;
    d->n_buckets = t->bucket_key.size();
    d->bucket_key = t->bucket_key.data();
    d->bucket_kmer_off = t->bucket_kmer_off.data();
    d->n_kmers = t->kmer_hash.size();
    d->kmer_hash = t->kmer_hash.data();
    d->kmer_node_off = t->kmer_node_off.data();
    d->node_ids = t->node_ids.data();
    return CLS_OK;
}

// Records are independent: contiguous slices on the host's threads; the pieces are in input order.
static void serialize_pieces(const cls_tree* t, const char* headers, const uint64_t* header_off, uint32_t n, const cls_placement* recs,
                             int format, std::vector<std::string>& o, std::vector<std::string>& e) {
    unsigned nt = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    if (n < 4096) nt = 1;
    o.assign(nt, std::string());
    e.assign(nt, std::string());
    std::vector<std::string> errs(nt);
    auto work = [&](unsigned w) {
        try {
            const uint32_t lo = (uint32_t)((uint64_t)n * w / nt), hi = (uint32_t)((uint64_t)n * (w + 1) / nt);
            o[w].reserve((size_t)(hi - lo) * 360);
            for (uint32_t i = lo; i < hi; ++i) {
                const std::string header(headers + header_off[i], headers + header_off[i + 1]);
                if (const char* et = error_text(recs[i].status)) { e[w] += et; continue; }  // mod.rs:160-169: appended without a newline
                serialize_one(t, header, recs[i], format, o[w]);
            }
        } catch (const std::exception& ex) { errs[w] = ex.what(); if (errs[w].empty()) errs[w] = "error"; }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned w = 0; w < nt; ++w) th.emplace_back(work, w);
        for (auto& x : th) x.join();
    }
    for (auto& m : errs) if (!m.empty()) throw std::runtime_error(m);
}

extern "C" int cls_serialize_results(const cls_tree* t, const char* headers, const uint64_t* header_off, uint32_t n,
                                     const cls_placement* recs, int format, char** out_text, size_t* out_len,
                                     char** err_text, size_t* err_len) {
    if (!t || !header_off || (!recs && n) || !out_text || !out_len) return fail(CLS_E_INVALID_ARG, "cls_serialize_results: null argument");
    try {
        std::vector<std::string> o, e;
        serialize_pieces(t, headers, header_off, n, recs, format, o, e);
        size_t ol = 0, el = 0;
        for (auto& x : o) ol += x.size();
        for (auto& x : e) el += x.size();
        *out_text = (char*)malloc(ol + 1);
        if (!*out_text) return fail(CLS_E_NOMEM, "cls_serialize_results: out of memory");
        { size_t p = 0; for (auto& x : o) { memcpy(*out_text + p, x.data(), x.size()); p += x.size(); } }
        (*out_text)[ol] = 0; *out_len = ol;
        if (err_text && err_len) {
            *err_text = (char*)malloc(el + 1);
            if (!*err_text) { free(*out_text); return fail(CLS_E_NOMEM, "cls_serialize_results: out of memory"); }
            { size_t p = 0; for (auto& x : e) { memcpy(*err_text + p, x.data(), x.size()); p += x.size(); } }
            (*err_text)[el] = 0; *err_len = el;
        }
        return CLS_OK;
    } catch (const std::exception& ex) {
        return fail(CLS_E_INTERNAL, std::string("cls_serialize_results: ") + ex.what());
    } catch (...) {
        return fail(CLS_E_INTERNAL, "cls_serialize_results: unknown exception");
    }
}

extern "C" void cls_host_free(void* p) { free(p); }

extern "C" int cls_place_sequences(cls_db* db, const cls_tree* t, const char* query_path, const char* out_file,
                                   const cls_params* params, int overwrite, int format, uint32_t* n_placed, double* seconds) {
    if (!db || !t || !query_path || !out_file) return fail(CLS_E_INVALID_ARG, "cls_place_sequences: null argument");
    // released on every way out, a throwing read_file / serialize_pieces included
    struct Guard {
        FILE *fo = nullptr, *fe = nullptr;
        cls_placement* recs = nullptr;
        cls_fasta fa{};
        ~Guard() { if (fo) fclose(fo); if (fe) fclose(fe); free(recs); cls_fasta_free(&fa); }
    } g;
    try {
        // ---- output paths + overwrite policy (mod.rs:73-106) ------------------------------------------
        const std::string out_path = with_extension(out_file, format == CLS_FORMAT_YAML ? "yaml" : "jsonl");
        const std::string err_path = with_extension(out_file, "error");
        size_t slash = out_path.find_last_of('/');
        if (slash != std::string::npos && slash > 0) (void)mkdir(out_path.substr(0, slash).c_str(), 0777);  // create_dir, error ignored
        struct stat sb;
        if (stat(out_path.c_str(), &sb) == 0) {
            if (!overwrite) return fail(CLS_E_INVALID_ARG, "Could not overwrite existing file \"" + out_path + "\" when overwrite option is `false`.");
            if (unlink(out_path.c_str()) != 0) return fail(CLS_E_INVALID_ARG, std::string("Could not remove file given ") + strerror(errno));
        }
        FILE*& fo = g.fo;
        FILE*& fe = g.fe;
        fo = fopen(out_path.c_str(), "ab");  // append mode, created even when nothing is written (write_or_append_to_file.rs:14-21)
        fe = fopen(err_path.c_str(), "ab");
        if (!fo || !fe) return fail(CLS_E_INVALID_ARG, "Unable to open file");
        // ---- read the WHOLE input first (mod.rs:118-119), then place, then write ---------------------------
        auto t0 = std::chrono::steady_clock::now();  // the reference's UCPLACE0001 -> UCPLACE0002 window (mod.rs:64-67, 264-267): read + place + write
        std::string text;
        if (strcmp(query_path, "-") == 0) { std::stringstream ss; ss << std::cin.rdbuf(); text = ss.str(); }
        else text = read_file(query_path);
        cls_fasta& fa = g.fa;
        cls_placement*& recs = g.recs;  // FASTA stage + placement on the device; headers + records come back
        auto t1 = std::chrono::steady_clock::now();
        int rc = cls_place_fasta_text(db, text.data(), text.size(), params, &fa, &recs);
        if (rc != CLS_OK) { std::string m = cls_last_error(); return fail(rc, m); }
        std::string().swap(text);
        const bool timing = cls::tuning().timing != 0;
        auto t2 = std::chrono::steady_clock::now();
        std::vector<std::string> po, pe;  // the pieces go to the files as they are: no second copy of 300 MB of text
        serialize_pieces(t, fa.headers, fa.header_off, fa.n, recs, format, po, pe);
        free(recs);
        recs = nullptr;
        auto t3 = std::chrono::steady_clock::now();
        for (auto& x : po) if (!x.empty() && fwrite(x.data(), 1, x.size(), fo) != x.size()) rc = fail(CLS_E_INTERNAL, "Error writing to file");
        for (auto& x : pe) if (!x.empty() && fwrite(x.data(), 1, x.size(), fe) != x.size()) rc = fail(CLS_E_INTERNAL, "Error writing to file");
        if (n_placed) *n_placed = fa.n;
        cls_fasta_free(&fa);
        fclose(fo); fclose(fe);
        fo = fe = nullptr;
        auto t4 = std::chrono::steady_clock::now();
        if (seconds) *seconds = std::chrono::duration<double>(t4 - t0).count();
        if (timing) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "cls_place_sequences: read %.1f ms, fasta+place %.1f ms, serialise %.1f ms, write %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
        }
        return rc;
    } catch (const std::exception& ex) {
        return fail(CLS_E_INTERNAL, std::string("cls_place_sequences: ") + ex.what());
    } catch (...) {
        return fail(CLS_E_INTERNAL, "cls_place_sequences: unknown exception");
    }
}
