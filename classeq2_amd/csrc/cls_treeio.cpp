// Tree / database files around the placement path:
//   Tree::init_from_file + get_children_nodes + sanitize + fix_parent_ids   core/src/domain/dtos/tree.rs:164-357
//   load_database (zstd-compressed YAML, plain YAML)                        ports/lib/src/functions/load_database.rs:9-53
//   `cls convert database -f {zstd,yaml,json} [--only-tree]`                 ports/cli/src/cmds/convert.rs:161-205
//   `cls build-db` output (zstd YAML)                                       ports/cli/src/cmds/build_db.rs:70-76
// in C++ (the reference is compiled Rust; no Rust toolchain in this image).  Newick parsing is the third-party
// `phylotree` crate in the reference (absent from /root/reference): node ids are its arena indices, i.e. the
// pre-order of the opening parentheses -- pinned by the reference's own Colletotrichum build
// (tests/golden/newick_colletotrichum.json).  YAML is read into the same DOM as JSON (cls_json.h), so one
// function (tree_from_doc) fills the tree from either.  zstd comes from the system's libzstd.so.1 at run time.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "cls_host.h"
#include "cls_host_internal.h"
#include "cls_json.h"

using namespace cls_host;

namespace {

// ---- YAML (the block-style subset serde_yaml writes) -> JVal ----------------------------------------------
class YParser {
public:
    explicit YParser(const std::string& text) {
        size_t i = 0;
        while (i <= text.size()) {
            size_t e = text.find('\n', i);
            if (e == std::string::npos) e = text.size();
            std::string ln = text.substr(i, e - i);
            if (!ln.empty() && ln.back() == '\r') ln.pop_back();
            lines_.push_back(std::move(ln));
            i = e + 1;
        }
    }
    cls::JVal parse() {
        skip_blank();
        if (pos_ < lines_.size() && lines_[pos_].compare(0, 3, "---") == 0) { ++pos_; skip_blank(); }
        if (pos_ >= lines_.size()) return cls::JVal();
        cls::JVal v = node(indent_of(lines_[pos_]));
        skip_blank();
        if (pos_ < lines_.size()) fail("unexpected content after the document");
        return v;
    }

private:
    std::vector<std::string> lines_;
    size_t pos_ = 0;
    [[noreturn]] void fail(const std::string& m) { throw std::runtime_error("yaml line " + std::to_string(pos_ + 1) + ": " + m); }
    static size_t indent_of(const std::string& s) { size_t i = 0; while (i < s.size() && s[i] == ' ') ++i; return i; }
    static bool blank(const std::string& s) { size_t i = indent_of(s); return i == s.size() || s[i] == '#'; }
    void skip_blank() { while (pos_ < lines_.size() && blank(lines_[pos_])) ++pos_; }
    static bool is_seq_item(const std::string& s, size_t ind) { return s.size() > ind && s[ind] == '-' && (s.size() == ind + 1 || s[ind + 1] == ' '); }

    // end of a mapping key that starts at s[i]; npos if the text is not "key:" / "key: value"
    static size_t key_end(const std::string& s, size_t i) {
        if (i >= s.size()) return std::string::npos;
        if (s[i] == '"' || s[i] == '\'') {
            const char q = s[i];
            size_t j = i + 1;
            while (j < s.size()) {
                if (q == '"' && s[j] == '\\') { j += 2; continue; }
                if (s[j] == q) { if (q == '\'' && j + 1 < s.size() && s[j + 1] == '\'') { j += 2; continue; } break; }
                ++j;
            }
            if (j >= s.size()) return std::string::npos;
            return (j + 1 < s.size() && s[j + 1] == ':' && (j + 2 == s.size() || s[j + 2] == ' ')) ? j + 1 : std::string::npos;
        }
        for (size_t j = i; j < s.size(); ++j)
            if (s[j] == ':' && (j + 1 == s.size() || s[j + 1] == ' ')) return j;
        return std::string::npos;
    }

    static std::string unquote(const std::string& s) {
        if (s.size() >= 2 && s.front() == '\'' && s.back() == '\'') {
            std::string o;
            for (size_t i = 1; i + 1 < s.size(); ++i) { if (s[i] == '\'' && s[i + 1] == '\'') ++i; o.push_back(s[i]); }
            return o;
        }
        if (s.size() >= 2 && s.front() == '"' && s.back() == '"') {
            std::string o;
            for (size_t i = 1; i + 1 < s.size(); ++i) {
                if (s[i] != '\\') { o.push_back(s[i]); continue; }
                const char c = s[++i];
                switch (c) {
                    case 'n': o.push_back('\n'); break;
                    case 't': o.push_back('\t'); break;
                    case 'r': o.push_back('\r'); break;
                    case '0': o.push_back('\0'); break;
                    case 'x': if (i + 2 < s.size()) { o.push_back((char)strtoul(s.substr(i + 1, 2).c_str(), nullptr, 16)); i += 2; } break;
                    case 'u': if (i + 4 < s.size()) {
                        const uint32_t cp = (uint32_t)strtoul(s.substr(i + 1, 4).c_str(), nullptr, 16);
                        i += 4;
                        if (cp < 0x80) o.push_back((char)cp);
                        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
                        else { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
                    } break;
                    default: o.push_back(c);
                }
            }
            return o;
        }
        return s;
    }

    static cls::JVal scalar(std::string s) {
        while (!s.empty() && (s.back() == ' ' || s.back() == '\t')) s.pop_back();
        cls::JVal v;
        if (!s.empty() && (s.front() == '"' || s.front() == '\'')) { v.kind = cls::JVal::Str; v.s = unquote(s); return v; }
        if (s.empty() || s == "~" || s == "null" || s == "Null" || s == "NULL") return v;
        if (s == "true" || s == "True" || s == "TRUE") { v.kind = cls::JVal::Bool; v.b = true; return v; }
        if (s == "false" || s == "False" || s == "FALSE") { v.kind = cls::JVal::Bool; return v; }
        if (s == "[]") { v.kind = cls::JVal::Arr; return v; }
        if (s == "{}") { v.kind = cls::JVal::Obj; return v; }
        char* end = nullptr;
        (void)strtod(s.c_str(), &end);
        const bool numeric = end && *end == '\0' && (isdigit((unsigned char)s[0]) || ((s[0] == '-' || s[0] == '+' || s[0] == '.') && s.size() > 1 && (isdigit((unsigned char)s[1]) || s[1] == '.')));
        v.kind = numeric ? cls::JVal::Num : cls::JVal::Str;
        v.s = s;
        return v;
    }

    // `| / |- / |+ / > ...` block scalar whose lines are indented deeper than `parent_ind`
    cls::JVal block_scalar(const std::string& header, size_t parent_ind) {
        const bool literal = header[0] == '|';
        const char chomp = header.size() > 1 ? header[1] : ' ';
        std::vector<std::string> body;
        size_t ind = std::string::npos;
        while (pos_ < lines_.size()) {
            const std::string& ln = lines_[pos_];
            const size_t i = indent_of(ln);
            if (i == ln.size()) { body.emplace_back(); ++pos_; continue; }  // empty line inside the block
            if (i <= parent_ind) break;
            if (ind == std::string::npos) ind = i;
            body.push_back(ln.substr(std::min(ind, ln.size())));
            ++pos_;
        }
        while (!body.empty() && body.back().empty() && chomp != '+') body.pop_back();
        std::string o;
        for (size_t i = 0; i < body.size(); ++i) {
            o += body[i];
            if (i + 1 < body.size()) o += literal ? "\n" : (body[i + 1].empty() || body[i].empty() ? "\n" : " ");
        }
        if (chomp != '-' && !body.empty()) o.push_back('\n');
        cls::JVal v;
        v.kind = cls::JVal::Str;
        v.s = o;
        return v;
    }

    // the value that follows "key:" or "- " : inline text `rest`, or nested lines deeper than `ind`
    cls::JVal value(const std::string& rest_in, size_t ind, bool in_mapping) {
        std::string rest = rest_in;
        while (!rest.empty() && rest.front() == ' ') rest.erase(rest.begin());
        if (!rest.empty() && rest[0] == '!') {  // `!Tag value`: serde's externally tagged enum -> {Tag: value}
            size_t sp = rest.find(' ');
            const std::string tag = rest.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
            cls::JVal inner = value(sp == std::string::npos ? "" : rest.substr(sp + 1), ind, in_mapping);
            cls::JVal v;
            v.kind = cls::JVal::Obj;
            v.obj.emplace_back(tag, std::move(inner));
            return v;
        }
        if (!rest.empty() && (rest[0] == '|' || rest[0] == '>') && (rest.size() == 1 || rest[1] == '-' || rest[1] == '+' || isdigit((unsigned char)rest[1]))) return block_scalar(rest, ind);
        if (!rest.empty()) return scalar(rest);
        skip_blank();
        if (pos_ >= lines_.size()) return cls::JVal();
        const size_t ni = indent_of(lines_[pos_]);
        if (ni > ind) return node(ni);
        // serde_yaml writes the items of a sequence held by a mapping key at the key's own indentation
        if (in_mapping && ni == ind && is_seq_item(lines_[pos_], ni)) return sequence(ni);
        return cls::JVal();
    }

    cls::JVal node(size_t ind) {
        skip_blank();
        if (pos_ >= lines_.size()) return cls::JVal();
        const std::string& ln = lines_[pos_];
        if (is_seq_item(ln, ind)) return sequence(ind);
        if (key_end(ln, ind) != std::string::npos) return mapping(ind);
        cls::JVal v = scalar(ln.substr(ind));
        ++pos_;
        return v;
    }

    cls::JVal sequence(size_t ind) {
        cls::JVal v;
        v.kind = cls::JVal::Arr;
        for (;;) {
            skip_blank();
            if (pos_ >= lines_.size() || indent_of(lines_[pos_]) != ind || !is_seq_item(lines_[pos_], ind)) break;
            std::string& ln = lines_[pos_];
            size_t c = ind + 1;
            while (c < ln.size() && ln[c] == ' ') ++c;
            if (c >= ln.size()) { ++pos_; v.arr.push_back(value("", ind, false)); continue; }
            if (ln[c] != '!' && ln[c] != '|' && ln[c] != '>' && key_end(ln, c) != std::string::npos) {
                // "- key: value": a mapping whose first key sits on the item line; its other keys are indented to `c`
                ln = std::string(c, ' ') + ln.substr(c);
                v.arr.push_back(mapping(c));
                continue;
            }
            const std::string rest = ln.substr(c);
            ++pos_;
            v.arr.push_back(value(rest, ind, false));
        }
        return v;
    }

    cls::JVal mapping(size_t ind) {
        cls::JVal v;
        v.kind = cls::JVal::Obj;
        for (;;) {
            skip_blank();
            if (pos_ >= lines_.size() || indent_of(lines_[pos_]) != ind) break;
            const std::string ln = lines_[pos_];
            if (is_seq_item(ln, ind)) break;
            const size_t ke = key_end(ln, ind);
            if (ke == std::string::npos) fail("expected `key:`");
            const std::string key = unquote(ln.substr(ind, ke - ind));
            ++pos_;
            v.obj.emplace_back(key, value(ke + 1 < ln.size() ? ln.substr(ke + 1) : "", ind, true));
        }
        return v;
    }
};

// ---- Newick ---------------------------------------------------------------------------------------------
// phylotree::Tree::from_newick: nodes are appended to the arena when they are opened, so ids follow the
// pre-order of the text; an internal node's label is its name (the reference parses it as the support
// value, tree.rs:339-345), `:x` is the length of the edge above the node.
struct NwkNode { std::string name; bool has_name = false, has_len = false; double len = 0; std::vector<uint32_t> children; };

class NewickParser {
public:
    explicit NewickParser(const std::string& s) : s_(s) {}
    std::vector<NwkNode> parse() {
        nodes_.clear();
        skip();
        (void)subtree();
        skip();
        if (i_ < s_.size() && s_[i_] == ';') ++i_;
        skip();
        if (i_ < s_.size()) throw std::runtime_error("newick: trailing characters at offset " + std::to_string(i_));
        return std::move(nodes_);
    }

private:
    const std::string& s_;
    size_t i_ = 0;
    std::vector<NwkNode> nodes_;
    void skip() {
        for (;;) {
            while (i_ < s_.size() && isspace((unsigned char)s_[i_])) ++i_;
            if (i_ < s_.size() && s_[i_] == '[') {  // comment
                int depth = 0;
                while (i_ < s_.size()) { if (s_[i_] == '[') ++depth; else if (s_[i_] == ']' && --depth == 0) { ++i_; break; } ++i_; }
                continue;
            }
            break;
        }
    }
    uint32_t subtree() {
        const uint32_t me = (uint32_t)nodes_.size();
        nodes_.emplace_back();
        skip();
        if (i_ < s_.size() && s_[i_] == '(') {
            ++i_;
            for (;;) {
                const uint32_t c = subtree();
                nodes_[me].children.push_back(c);
                skip();
                if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
                if (i_ < s_.size() && s_[i_] == ')') { ++i_; break; }
                throw std::runtime_error("newick: expected ',' or ')' at offset " + std::to_string(i_));
            }
        }
        skip();
        std::string label;
        if (i_ < s_.size() && s_[i_] == '\'') {
            ++i_;
            while (i_ < s_.size()) {
                if (s_[i_] == '\'') { if (i_ + 1 < s_.size() && s_[i_ + 1] == '\'') { label.push_back('\''); i_ += 2; continue; } ++i_; break; }
                label.push_back(s_[i_++]);
            }
            nodes_[me].has_name = true;
        } else {
            while (i_ < s_.size() && !strchr("(),:;[", s_[i_]) && !isspace((unsigned char)s_[i_])) label.push_back(s_[i_++]);
            nodes_[me].has_name = !label.empty();
        }
        nodes_[me].name = label;
        skip();
        if (i_ < s_.size() && s_[i_] == ':') {
            ++i_;
            skip();
            const char* b = s_.c_str() + i_;
            char* e = nullptr;
            const double v = strtod(b, &e);
            if (e == b) throw std::runtime_error("newick: bad branch length at offset " + std::to_string(i_));
            i_ += (size_t)(e - b);
            nodes_[me].has_len = true;
            nodes_[me].len = v;
        }
        return me;
    }
};

// Tree::get_children_nodes (tree.rs:293-357)
std::vector<Clade> children_of(const std::vector<NwkNode>& nodes, uint32_t id) {
    std::vector<Clade> out;
    for (uint32_t c : nodes[id].children) {
        const NwkNode& n = nodes[c];
        Clade k;
        k.id = c;
        k.has_parent = true;
        k.parent = id;
        k.has_length = n.has_len;
        k.length = n.len;
        if (n.children.empty()) {  // Clade::new_leaf
            k.kind = CLS_KIND_LEAF;
            k.has_name = true;
            k.name = n.has_name ? n.name : "Unnamed";
        } else {  // Clade::new_internal: no name; support = the label if it parses as f64
            k.kind = CLS_KIND_NODE;
            if (n.has_name) {
                char* e = nullptr;
                const double v = strtod(n.name.c_str(), &e);
                if (e != n.name.c_str() && *e == '\0') { k.has_support = true; k.support = v; }
            }
            k.has_children = true;
            k.children = children_of(nodes, c);
        }
        out.push_back(std::move(k));
    }
    return out;
}

// Tree::sanitize (tree.rs:252-291): a child whose support is below the minimum gives its children to its parent
void sanitize(Clade& clade, double min_support) {
    std::vector<Clade> kept;
    for (Clade& child : clade.children) {
        sanitize(child, min_support);
        if (!child.has_support || child.support >= min_support || child.kind == CLS_KIND_LEAF) kept.push_back(std::move(child));
        else for (Clade& g : child.children) kept.push_back(std::move(g));
    }
    clade.children = std::move(kept);
    clade.has_children = !clade.children.empty();
}

void fix_parent_ids(Clade& c) {  // tree.rs:232-249
    for (Clade& ch : c.children) { ch.has_parent = true; ch.parent = c.id; fix_parent_ids(ch); }
}

// ---- MD5 / UUID v3 (Tree.id = Uuid::new_v3(NAMESPACE_DNS, file name), tree.rs:222) -------------------------------
struct Md5 {
    uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476;
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
            0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
            0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
            0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
            0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
            0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int R[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f;
            int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D;
            D = C;
            C = B;
            const uint32_t x = A + f + K[i] + w[g];
            B = B + ((x << R[i]) | (x >> (32 - R[i])));
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    static void digest(const std::string& msg, uint8_t out[16]) {
        Md5 m;
        std::string s = msg;
        const uint64_t bits = (uint64_t)msg.size() * 8;
        s.push_back((char)0x80);
        while (s.size() % 64 != 56) s.push_back('\0');
        for (int i = 0; i < 8; ++i) s.push_back((char)(bits >> (8 * i)));
        for (size_t i = 0; i < s.size(); i += 64) m.block(reinterpret_cast<const uint8_t*>(s.data()) + i);
        const uint32_t v[4] = {m.a, m.b, m.c, m.d};
        for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(v[i / 4] >> (8 * (i % 4)));
    }
};

std::string uuid_v3_dns(const std::string& name) {
    static const uint8_t NS_DNS[16] = {0x6b, 0xa7, 0xb8, 0x10, 0x9d, 0xad, 0x11, 0xd1, 0x80, 0xb4, 0x00, 0xc0, 0x4f, 0xd4, 0x30, 0xc8};
    uint8_t h[16];
    Md5::digest(std::string(reinterpret_cast<const char*>(NS_DNS), 16) + name, h);
    h[6] = (uint8_t)((h[6] & 0x0F) | 0x30);
    h[8] = (uint8_t)((h[8] & 0x3F) | 0x80);
    char buf[40];
    snprintf(buf, sizeof buf, "%02x%02x%02x%02x-%02x%02x-%02x%02x-%02x%02x-%02x%02x%02x%02x%02x%02x", h[0], h[1], h[2], h[3], h[4], h[5], h[6],
             h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
    return buf;
}

// ---- zstd through the system library (no headers in this image: the stable one-shot / streaming ABI) ------------
struct ZIn { const void* src; size_t size; size_t pos; };
struct ZOut { void* dst; size_t size; size_t pos; };
struct Zstd {
    void* lib = nullptr;
    void* (*createDStream)() = nullptr;
    size_t (*freeDStream)(void*) = nullptr;
    size_t (*decompressStream)(void*, ZOut*, ZIn*) = nullptr;
    size_t (*compressBound)(size_t) = nullptr;
    size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char* (*getErrorName)(size_t) = nullptr;
    static Zstd& get() {
        static Zstd z = [] {
            Zstd r;
            for (const char* n : {"libzstd.so.1", "libzstd.so"}) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
            if (!r.lib) return r;
            r.createDStream = (void* (*)())dlsym(r.lib, "ZSTD_createDStream");
            r.freeDStream = (size_t(*)(void*))dlsym(r.lib, "ZSTD_freeDStream");
            r.decompressStream = (size_t(*)(void*, ZOut*, ZIn*))dlsym(r.lib, "ZSTD_decompressStream");
            r.compressBound = (size_t(*)(size_t))dlsym(r.lib, "ZSTD_compressBound");
            r.compress = (size_t(*)(void*, size_t, const void*, size_t, int))dlsym(r.lib, "ZSTD_compress");
            r.isError = (unsigned (*)(size_t))dlsym(r.lib, "ZSTD_isError");
            r.getErrorName = (const char* (*)(size_t))dlsym(r.lib, "ZSTD_getErrorName");
            if (!r.createDStream || !r.freeDStream || !r.decompressStream || !r.compressBound || !r.compress || !r.isError) r.lib = nullptr;
            return r;
        }();
        return z;
    }
};

bool is_zstd(const std::string& s) { return s.size() >= 4 && (uint8_t)s[0] == 0x28 && (uint8_t)s[1] == 0xB5 && (uint8_t)s[2] == 0x2F && (uint8_t)s[3] == 0xFD; }

std::string zstd_decompress(const std::string& in) {
    Zstd& z = Zstd::get();
    if (!z.lib) throw std::runtime_error("libzstd.so.1 is not available: cannot read a zstd-compressed database");
    void* ds = z.createDStream();
    if (!ds) throw std::runtime_error("ZSTD_createDStream failed");
    std::string out;
    std::vector<char> buf(1 << 20);
    ZIn zi{in.data(), in.size(), 0};
    size_t rc = 1;
    while (zi.pos < zi.size || rc != 0) {
        ZOut zo{buf.data(), buf.size(), 0};
        rc = z.decompressStream(ds, &zo, &zi);
        if (z.isError(rc)) { z.freeDStream(ds); throw std::runtime_error(std::string("zstd: ") + (z.getErrorName ? z.getErrorName(rc) : "decompression failed")); }
        out.append(buf.data(), zo.pos);
        if (zi.pos >= zi.size && zo.pos == 0) break;  // input exhausted and nothing more came out
    }
    z.freeDStream(ds);
    return out;
}

std::string zstd_compress(const std::string& in) {
    Zstd& z = Zstd::get();
    if (!z.lib) throw std::runtime_error("libzstd.so.1 is not available: cannot write a zstd-compressed database");
    std::string out(z.compressBound(in.size()), '\0');
    const size_t n = z.compress(&out[0], out.size(), in.data(), in.size(), 3);  // zstd::Encoder::new(w, 0) = the default level
    if (z.isError(n)) throw std::runtime_error("zstd: compression failed");
    out.resize(n);
    return out;
}

// ---- writers: serde_yaml::to_writer / serde_json::to_writer_pretty of Tree or Clade ----------------------------
void yaml_annotations(std::string& o, const cls_tree* t) {
    o += "annotations:\n";
    for (auto& a : t->annotations) {
        o += "- clade: " + std::to_string(a.clade) + "\n";
        if (a.has_meta) {
            if (a.meta.empty()) { o += "  meta: []\n"; continue; }
            o += "  meta:\n";
            for (auto& g : a.meta) {
                o += "  - !" + g.name + " ";
                if (g.is_int) o += std::to_string(g.ival) + "\n";
                else yaml_str(o, g.sval, 4);
            }
        }
    }
}

std::string tree_yaml(const cls_tree* t, bool only_tree) {
    std::string o;
    if (only_tree || !t->has_header) { yaml_clade(o, t->root, 0, false); return o; }
    o += "id: " + t->uuid + "\n";
    o += "name: ";
    yaml_str(o, t->name, 0);
    o += "minBranchSupport: " + fmt_f64(t->min_branch_support) + "\n";
    o += "inMemorySize: ";
    if (t->has_in_memory_size) yaml_str(o, t->in_memory_size, 0); else o += "null\n";
    o += "root:\n";
    yaml_clade(o, t->root, 2, false);
    if (t->has_annotations) yaml_annotations(o, t);
    if (!t->has_kmers) { o += "kmersMap: null\n"; return o; }
    o += "kmersMap:\n  kSize: " + std::to_string(t->k_size) + "\n  mSize: " + std::to_string(t->m_size) + "\n";
    if (t->bucket_key.empty()) { o += "  map: {}\n"; return o; }
    o += "  map:\n";
    for (size_t b = 0; b < t->bucket_key.size(); ++b) {
        o += "    " + std::to_string(t->bucket_key[b]) + ":";
        if (t->bucket_kmer_off[b] == t->bucket_kmer_off[b + 1]) { o += " {}\n"; continue; }
        o += "\n";
        for (uint64_t j = t->bucket_kmer_off[b]; j < t->bucket_kmer_off[b + 1]; ++j) {
            o += "      " + std::to_string(t->kmer_hash[j]) + ":";
            if (t->kmer_node_off[j] == t->kmer_node_off[j + 1]) { o += " []\n"; continue; }
            o += "\n";
            for (uint64_t i = t->kmer_node_off[j]; i < t->kmer_node_off[j + 1]; ++i) o += "      - " + std::to_string(t->node_ids[i]) + "\n";
        }
    }
    return o;
}

void json_clade_pretty(std::string& o, const Clade& c, size_t ind) {
    const std::string in1(ind + 2, ' ');
    o += "{\n";
    o += in1 + "\"id\": " + std::to_string(c.id) + ",\n";
    o += in1 + "\"parent\": " + (c.has_parent ? std::to_string(c.parent) : "null") + ",\n";
    o += in1 + "\"kind\": \"" + (c.kind == CLS_KIND_ROOT ? "ROOT" : c.kind == CLS_KIND_LEAF ? "LEAF" : "NODE") + "\"";
    if (c.has_name) { o += ",\n" + in1 + "\"name\": "; json_str(o, c.name); }
    if (c.has_support) o += ",\n" + in1 + "\"support\": " + json_f64(c.support);
    if (c.has_length) o += ",\n" + in1 + "\"length\": " + json_f64(c.length);
    if (c.has_children) {
        o += ",\n" + in1 + "\"children\": [";
        if (c.children.empty()) o += "]";
        else {
            for (size_t i = 0; i < c.children.size(); ++i) {
                o += i ? ",\n" : "\n";
                o += std::string(ind + 4, ' ');
                json_clade_pretty(o, c.children[i], ind + 4);
            }
            o += "\n" + in1 + "]";
        }
    }
    o += "\n" + std::string(ind, ' ') + "}";
}

std::string tree_json(const cls_tree* t, bool only_tree) {
    std::string o;
    if (only_tree || !t->has_header) { json_clade_pretty(o, t->root, 0); return o; }
    o += "{\n  \"id\": ";
    json_str(o, t->uuid);
    o += ",\n  \"name\": ";
    json_str(o, t->name);
    o += ",\n  \"minBranchSupport\": " + json_f64(t->min_branch_support);
    o += ",\n  \"inMemorySize\": ";
    if (t->has_in_memory_size) json_str(o, t->in_memory_size); else o += "null";
    o += ",\n  \"root\": ";
    json_clade_pretty(o, t->root, 2);
    if (t->has_annotations) {
        o += ",\n  \"annotations\": [";
        for (size_t i = 0; i < t->annotations.size(); ++i) {
            const Annotation& a = t->annotations[i];
            o += i ? ",\n    {\n" : "\n    {\n";
            o += "      \"clade\": " + std::to_string(a.clade) + ",\n      \"meta\": ";
            if (!a.has_meta) o += "null";
            else if (a.meta.empty()) o += "[]";
            else {
                o += "[";
                for (size_t g = 0; g < a.meta.size(); ++g) {
                    o += g ? ",\n        {\n          " : "\n        {\n          ";
                    json_str(o, a.meta[g].name);
                    o += ": ";
                    if (a.meta[g].is_int) o += std::to_string(a.meta[g].ival); else json_str(o, a.meta[g].sval);
                    o += "\n        }";
                }
                o += "\n      ]";
            }
            o += "\n    }";
        }
        o += t->annotations.empty() ? "]" : "\n  ]";
    }
    o += ",\n  \"kmersMap\": ";
    if (!t->has_kmers) { o += "null\n}"; return o; }
    o += "{\n    \"kSize\": " + std::to_string(t->k_size) + ",\n    \"mSize\": " + std::to_string(t->m_size) + ",\n    \"map\": {";
    for (size_t b = 0; b < t->bucket_key.size(); ++b) {
        o += b ? ",\n      \"" : "\n      \"";
        o += std::to_string(t->bucket_key[b]) + "\": {";
        for (uint64_t j = t->bucket_kmer_off[b]; j < t->bucket_kmer_off[b + 1]; ++j) {
            o += j > t->bucket_kmer_off[b] ? ",\n        \"" : "\n        \"";
            o += std::to_string(t->kmer_hash[j]) + "\": [";
            for (uint64_t i = t->kmer_node_off[j]; i < t->kmer_node_off[j + 1]; ++i) {
                o += i > t->kmer_node_off[j] ? ",\n          " : "\n          ";
                o += std::to_string(t->node_ids[i]);
            }
            o += t->kmer_node_off[j] == t->kmer_node_off[j + 1] ? "]" : "\n        ]";
        }
        o += t->bucket_kmer_off[b] == t->bucket_kmer_off[b + 1] ? "}" : "\n      }";
    }
    o += t->bucket_key.empty() ? "}" : "\n    }";
    o += "\n  }\n}";
    return o;
}

std::string file_name_of(const std::string& path) { size_t s = path.find_last_of('/'); return s == std::string::npos ? path : path.substr(s + 1); }

}  // namespace

extern "C" int cls_tree_from_newick(const char* newick_text, const char* tree_name, double min_branch_support, cls_tree** out) {
    if (!newick_text || !out) return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_from_newick: null argument");
    try {
        const std::string text(newick_text);
        std::vector<NwkNode> nodes = NewickParser(text).parse();
        if (nodes.empty()) return cls_host_fail(CLS_E_BAD_TREE, "cls_tree_from_newick: empty tree");
        // "Tree is not rooted" (tree.rs:201-203) asks phylotree's is_rooted(); the reference's own unit test
        // (tree.rs:366-376) feeds it the Colletotrichum tree, whose root has THREE children, and expects Ok: the
        // check does not look at the root's arity.  A text without any clade below the root is refused here.
        if (nodes[0].children.empty()) return cls_host_fail(CLS_E_BAD_TREE, "cls_tree_from_newick: Tree is not rooted");
        auto t = std::make_unique<cls_tree>();
        Clade& root = t->root;  // Clade::new_root(0.0, children)
        root.id = 0;
        root.kind = CLS_KIND_ROOT;
        root.has_length = true;
        root.length = 0.0;
        root.children = children_of(nodes, 0);
        root.has_children = true;
        sanitize(root, min_branch_support);
        fix_parent_ids(root);
        root.has_parent = false;
        t->has_header = true;
        t->name = tree_name && *tree_name ? tree_name : "UnnamedTree";
        t->uuid = uuid_v3_dns(t->name);
        t->min_branch_support = min_branch_support;
        flatten(t.get());
        *out = t.release();
        return CLS_OK;
    } catch (const std::exception& e) {
        return cls_host_fail(CLS_E_BAD_TREE, std::string("cls_tree_from_newick: ") + e.what());
    } catch (...) {
        return cls_host_fail(CLS_E_INTERNAL, "cls_tree_from_newick: unknown exception");
    }
}

extern "C" int cls_tree_init_from_file(const char* tree_path, double min_branch_support, cls_tree** out) {
    if (!tree_path || !out) return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_init_from_file: null argument");
    try {
        const std::string path(tree_path);
        const size_t dot = path.find_last_of('.');
        const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
        if (ext != "nwk" && ext != "newick" && ext != "tree") return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_init_from_file: Tree file format is not supported");  // tree.rs:168-177
        return cls_tree_from_newick(read_file(tree_path).c_str(), file_name_of(path).c_str(), min_branch_support, out);
    } catch (const std::exception& e) {
        return cls_host_fail(CLS_E_BAD_TREE, std::string("cls_tree_init_from_file: ") + e.what());
    }
}

extern "C" int cls_tree_load(const char* path, cls_tree** out) {
    if (!path || !out) return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_load: null argument");
    try {
        std::string text = read_file(path);
        if (is_zstd(text)) text = zstd_decompress(text);
        size_t i = 0;
        while (i < text.size() && isspace((unsigned char)text[i])) ++i;
        cls::JVal doc = (i < text.size() && text[i] == '{') ? cls::JParser(text.data(), text.size()).parse() : YParser(text).parse();
        if (doc.kind != cls::JVal::Obj) return cls_host_fail(CLS_E_BAD_DB, "cls_tree_load: the file does not hold a database or tree mapping");
        auto t = std::make_unique<cls_tree>();
        tree_from_doc(doc, t.get());
        *out = t.release();
        return CLS_OK;
    } catch (const std::exception& e) {
        return cls_host_fail(CLS_E_BAD_DB, std::string("cls_tree_load: ") + e.what());
    } catch (...) {
        return cls_host_fail(CLS_E_INTERNAL, "cls_tree_load: unknown exception");
    }
}

extern "C" int cls_tree_serialize(const cls_tree* t, int format, int only_tree, char** out, size_t* out_len) {
    if (!t || !out || !out_len) return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_serialize: null argument");
    try {
        std::string s;
        if (format == CLS_DB_FORMAT_JSON) s = tree_json(t, only_tree != 0);
        else if (format == CLS_DB_FORMAT_YAML) s = tree_yaml(t, only_tree != 0);
        else if (format == CLS_DB_FORMAT_ZSTD) s = zstd_compress(tree_yaml(t, only_tree != 0));
        else return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_serialize: unknown format");
        char* p = (char*)malloc(s.size() + 1);
        if (!p) return cls_host_fail(CLS_E_NOMEM, "cls_tree_serialize: out of memory");
        memcpy(p, s.data(), s.size());
        p[s.size()] = '\0';
        *out = p;
        *out_len = s.size();
        return CLS_OK;
    } catch (const std::exception& e) {
        return cls_host_fail(CLS_E_INTERNAL, std::string("cls_tree_serialize: ") + e.what());
    }
}

extern "C" int cls_tree_save(const cls_tree* t, const char* path, int format, int only_tree) {
    if (!t || !path) return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_save: null argument");
    char* buf = nullptr;
    size_t n = 0;
    int rc = cls_tree_serialize(t, format, only_tree, &buf, &n);
    if (rc != CLS_OK) return rc;
    // the extension the reference's commands force (convert.rs:171,184,195; build_db.rs:72)
    const std::string dst = with_extension(path, format == CLS_DB_FORMAT_JSON ? "cls.json" : format == CLS_DB_FORMAT_YAML ? "cls.yaml" : "cls");
    FILE* f = fopen(dst.c_str(), "wb");
    if (!f) { free(buf); return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_save: cannot create " + dst); }
    const bool ok = fwrite(buf, 1, n, f) == n;
    free(buf);
    if (fclose(f) != 0 || !ok) return cls_host_fail(CLS_E_INTERNAL, "cls_tree_save: write failed: " + dst);
    return CLS_OK;
}
