// cls-place: C++ look-alike of the reference's `cls place` sub-command
// (ports/cli/src/cmds/place_sequences.rs:18-82 flag surface, :84-223 behaviour) on the GPU path.
//   cls-place [QUERY|-] -d DB -o OUT [-a ANNOTATIONS.yaml] [--out-format yaml|jsonl]
//             [-i N] [-m COV] [-r] [-f] [--device N]
// The database is read like load_database does (ports/lib/src/functions/load_database.rs:9-53): the `.cls`
// file of `cls build-db` (zstd-compressed YAML), plain YAML, or the JSON export.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "cls_host.h"

static void usage() {
    fprintf(stderr,
            "Usage: cls-place [QUERY] --database-file-path <DB> --output-file-path <OUT> [OPTIONS]\n\n"
            "Arguments:\n  [QUERY]  multi-FASTA file, or \"-\" for STDIN [default: -]\n\n"
            "Options:\n"
            "  -d, --database-file-path <PATH>     classeq database (.cls, .cls.yaml or .cls.json)\n"
            "  -o, --output-file-path <PATH>       output file (extension replaced by .yaml / .jsonl; errors go to .error)\n"
            "  -a, --annotations-file-path <PATH>  annotations in YAML format\n"
            "      --out-format <yaml|jsonl>       [default: yaml]\n"
            "  -i, --iterations <N>                maximum number of tree levels [default: 1000]\n"
            "  -m, --match-coverage <F>            minimum match coverage [default: 0.7]\n"
            "  -r, --remove-intersection           one-vs-rest without the shared k-mers\n"
            "  -f, --force-overwrite               overwrite an existing output file\n"
            "      --device <N>                    GPU ordinal [default: 0]\n");
}

int main(int argc, char** argv) {
    std::string query = "-", db_path, out_path, ann_path, fmt = "yaml";
    cls_params p;
    memset(&p, 0, sizeof p);
    int overwrite = 0, device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "error: a value is required for '%s'\n", name); exit(2); }
            return argv[++i];
        };
        if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a == "-d" || a == "--database-file-path") db_path = need("--database-file-path");
        else if (a == "-o" || a == "--output-file-path") out_path = need("--output-file-path");
        else if (a == "-a" || a == "--annotations-file-path") ann_path = need("--annotations-file-path");
        else if (a == "--out-format") fmt = need("--out-format");
        else if (a == "-i" || a == "--iterations") { p.flags |= CLS_HAS_MAX_ITERATIONS; p.max_iterations = atoi(need("--iterations")); }
        else if (a == "-m" || a == "--match-coverage") { p.flags |= CLS_HAS_MIN_MATCH_COVERAGE; p.min_match_coverage = atof(need("--match-coverage")); }
        else if (a == "-r" || a == "--remove-intersection") { p.flags |= CLS_HAS_REMOVE_INTERSECTION; p.remove_intersection = 1; }
        else if (a == "-f" || a == "--force-overwrite") overwrite = 1;
        else if (a == "--device") device = atoi(need("--device"));
        else if (!a.empty() && a[0] == '-' && a != "-") { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); usage(); return 2; }
        else query = a;
    }
    if (db_path.empty() || out_path.empty()) { usage(); return 2; }
    if (fmt != "yaml" && fmt != "jsonl") { fprintf(stderr, "error: invalid value '%s' for '--out-format'\n", fmt.c_str()); return 2; }

    cls_tree* tree = nullptr;
    if (cls_tree_load(db_path.c_str(), &tree) != CLS_OK) { fprintf(stderr, "Error loading database: %s\n", cls_host_last_error()); return 1; }
    if (!ann_path.empty() && cls_tree_set_annotations_yaml(tree, ann_path.c_str()) != CLS_OK) {
        fprintf(stderr, "Error loading annotations: %s\n", cls_host_last_error());
        return 1;
    }
    cls_db_desc desc;
    cls_db* db = nullptr;
    if (cls_tree_desc(tree, &desc) != CLS_OK) { fprintf(stderr, "%s\n", cls_host_last_error()); return 1; }
    if (cls_db_create(&desc, device, &db) != CLS_OK) { fprintf(stderr, "%s\n", cls_last_error()); return 1; }
    uint32_t n = 0;
    double seconds = 0;
    int rc = cls_place_sequences(db, tree, query.c_str(), out_path.c_str(), &p, overwrite, fmt == "yaml" ? CLS_FORMAT_YAML : CLS_FORMAT_JSONL, &n, &seconds);
    if (rc != CLS_OK) fprintf(stderr, "%s\n", cls_host_last_error());
    else fprintf(stderr, "{\"code\":\"CLIPLACE0002\",\"sequences\":%u,\"totalSeconds\":%.6f,\"averageSeconds\":%.9f}\n", n, seconds, n ? seconds / n : 0.0);
    cls_db_destroy(db);
    cls_tree_free(tree);
    return rc == CLS_OK ? 0 : 1;
}
