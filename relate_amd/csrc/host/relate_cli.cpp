// relateHMM_hip -- host driver over the C ABI (include/relate_hip.h): keeps the reference CLI's flags, its
// 27-line option file, .geno/.pos/.chr/.keep text inputs, PLINK .bed/.bim/.fam inputs and the .post/.k/stripped
// output files (src/runRelate.cpp:163-485, src/filereader_and_conversions.cpp, src/asort.cpp,
// src/relateHMM.cpp:117-175, 3739-3790, 3830-3880), and replaces the serial pair loop with batched
// relate_hip_run_pairs calls.  Host C++ only: every number written comes from the library's HIP kernels.
//
// Differences from the reference, on purpose:
//   * a lone pair always goes through init_globals + sgl (the reference's `-choose 0`; its default slow path
//     getPars/single_pair is the superseded twin);
//   * `-seed N` pins the start-point stream (the reference seeds with time(0) per pair, :1853);
//   * `-gpu D` selects the device; errors exit with status 1 instead of the reference's exit(0).
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <string>
#include <vector>

#include "relate_hip.h"

namespace {

const char *kDelims = ",;: \t\r";

[[noreturn]] void die(const std::string &msg) {
  std::fprintf(stderr, "\t-> %s\n", msg.c_str());
  std::exit(1);
}

bool file_exists(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}

// update_filename (src/relateHMM.cpp:157-175): first of name, name1, name2, ... that does not exist
std::string fresh_name(const std::string &name) {
  if (!file_exists(name)) return name;
  for (int i = 1;; i++) {
    std::string cand = name + std::to_string(i);
    if (!file_exists(cand)) {
      std::printf("\t-> Filename: %s exists will instead use filename:%s\n", name.c_str(), cand.c_str());
      return cand;
    }
  }
}

// tokens of a whole text file (readarray, src/filereader_and_conversions.cpp:230-257); rows are returned too
// so that a matrix reader can check row lengths (readmatrix, :317-417)
void read_tokens(const std::string &path, std::vector<std::string> *tokens, std::vector<size_t> *row_len) {
  std::ifstream in(path);
  if (!in) die("Problems opening file: " + path + "\t right directory?");
  std::string line;
  while (std::getline(in, line)) {
    size_t before = tokens->size();
    size_t i = 0;
    while (i < line.size()) {
      size_t b = line.find_first_not_of(kDelims, i);
      if (b == std::string::npos) break;
      size_t e = line.find_first_of(kDelims, b);
      if (e == std::string::npos) e = line.size();
      tokens->emplace_back(line, b, e - b);
      i = e;
    }
    if (row_len && tokens->size() > before) row_len->push_back(tokens->size() - before);
  }
}

int token_to_int(const std::string &t) {  // to_int, :118-151: NA -> 0
  if (t == "NA" || t == "Na" || t == "na") return 0;
  char *end = nullptr;
  long v = std::strtol(t.c_str(), &end, 10);
  if (end == t.c_str()) die("Error converting :\"" + t + "\" will exit");
  return (int)v;
}

double token_to_double(const std::string &t) {
  char *end = nullptr;
  double v = std::strtod(t.c_str(), &end);
  if (end == t.c_str()) die("Error converting :\"" + t + "\" will exit");
  return v;
}

struct DataSet {
  int N = 0, M = 0;
  std::vector<uint8_t> geno;  // N x M
  std::vector<uint8_t> bed;   // SNP-major PLINK payload (then geno stays empty and cols names its rows)
  std::vector<int> cols;
  int bed_rows = 0;
  std::vector<double> pos;
  std::vector<int> chr;
};

// readmatrix (:317-417) for the genotype file, streamed line by line into the byte matrix (a token vector of a
// 2 000 x 500 000 file would not fit in memory)
void read_geno_text(const std::string &path, DataSet *d) {
  std::ifstream in(path);
  if (!in) die("Problems opening file: " + path + "\t right directory?");
  std::string line, tok;
  size_t first = 0, nrows = 0;
  while (std::getline(in, line)) {
    size_t n = 0, i = 0;
    while (i < line.size()) {
      size_t b = line.find_first_not_of(kDelims, i);
      if (b == std::string::npos) break;
      size_t e = line.find_first_of(kDelims, b);
      if (e == std::string::npos) e = line.size();
      int v;
      if (e - b == 1 && line[b] >= '0' && line[b] <= '9') v = line[b] - '0';
      else { tok.assign(line, b, e - b); v = token_to_int(tok); }
      if (v < 0 || v > 3)
        die("Error in genotype data: (" + std::to_string(nrows) + "," + std::to_string(n) + ")=" + std::to_string(v) +
            ", value should be between 0 and 3.");
      d->geno.push_back((uint8_t)v);
      n++;
      i = e;
    }
    if (!n) continue;  // blank line
    if (!nrows) { first = n; d->geno.reserve(first * 128); }
    else if (n != first)
      die("row length mismatch at line:" + std::to_string(nrows + 1) + " numitems is:" + std::to_string(first) +
          " shouldn't be:" + std::to_string(n) + "\t will exit");
    nrows++;
  }
  if (!nrows) die("genotype file is empty: " + path);
  d->N = (int)nrows;
  d->M = (int)first;
}

// 27-line option file: first tab-separated token of each line through atof into a float
// (src/filereader_and_conversions.cpp:31-94), then post_processing_of_option_pars (src/runRelate.cpp:76-101)
struct Options {
  relate_opts o{};
  int doAllPairs = 0;
  int pair[2] = {0, 1};
};

void set_defaults(Options *op) {  // setDefaultParsRun, src/runRelate.cpp:104-147
  relate_opts &o = op->o;
  o.alim[0] = 0.001; o.alim[1] = 0.15;
  o.min = 0; o.double_recom = 0; o.LD = 1; o.ld_adj = 1; o.epsilon = 0.01; o.back = 5; o.doPrune = 0;
  o.fixA = 0; o.fixK = 1; o.calcA = 1; o.fixA_val = 0; o.fixK_val = 0; o.prune_val = 0.2; o.phi = 0.013;
  o.convTol = 0.1; o.timesToConverge = 2; o.timesToRun = 3; o.doPar = 0;
}

void read_options(const std::string &path, Options *op) {
  std::ifstream in(path);
  if (!in) die("Error opening options file");
  float v[27];
  int n = 0;
  std::string line;
  while (n < 27 && std::getline(in, line)) {
    size_t b = line.find_first_not_of('\t');
    std::string first = b == std::string::npos ? "" : line.substr(b, line.find('\t', b) - b);
    v[n++] = (float)std::atof(first.c_str());
  }
  if (n < 27) die("options file has fewer than 27 lines: " + path);
  relate_opts &o = op->o;
  op->doAllPairs = (int)v[0];
  op->pair[0] = (int)v[1]; op->pair[1] = (int)v[2];
  o.double_recom = (int)v[3];
  o.LD = ((int)v[4] == 0) ? 1 : 0;  // inverted on purpose (:69): 0 in the file = r^2
  o.min = (double)v[5];
  o.alim[0] = v[6]; o.alim[1] = v[7];
  o.doPar = (int)v[8];
  o.par[0] = v[9]; o.par[1] = v[10]; o.par[2] = v[11];
  o.ld_adj = (int)v[12];
  o.epsilon = v[13];
  o.back = (int)v[14];
  o.doPrune = (int)v[15];
  o.prune_val = v[16];
  o.fixA = (int)v[17]; o.fixA_val = v[18];
  o.fixK = (int)v[19]; o.fixK_val = v[20];
  o.calcA = (int)v[21];
  o.phi = v[22];
  o.convTol = v[23];
  o.timesToConverge = (int)v[24];
  o.timesToRun = (int)v[25];
  // line 27 (back2) is overwritten with 2*back by the reference (src/runRelate.cpp:88); the library does the same
  if (o.calcA && o.fixK && o.fixK_val == 0 && o.timesToRun != 1) {
    std::printf("\t-> Calc.A and fixK=0 is a one-dim optimization\n\t-> So will only need to run optimization once.\n");
    o.timesToRun = 1;
  }
  if (o.doPrune && o.back == 0) die("Must supply a back > 0, when you want to prune");
}

// PLINK: doBimFile (autosomes 1..22, position / 1e6) + bed_to_iMatrix (recode, 1 <-> 3 swap)
// (src/filereader_and_conversions.cpp:483-657)
void read_plink(const std::string &bed, const std::string &bim, const std::string &fam, DataSet *d) {
  std::vector<int> chr_all;
  std::vector<double> pos_all;
  {
    std::ifstream in(bim);
    if (!in) die("Problems opening file: " + bim + "\t right directory?");
    std::string line;
    int row = 0;
    while (std::getline(in, line)) {
      std::vector<std::string> t;
      size_t i = 0;
      while (i < line.size()) {
        size_t b = line.find_first_not_of(" \t\r", i);
        if (b == std::string::npos) break;
        size_t e = line.find_first_of(" \t\r", b);
        if (e == std::string::npos) e = line.size();
        t.emplace_back(line, b, e - b);
        i = e;
      }
      if (t.empty()) break;
      if (t.size() != 6) die("plink bim file:" + bim + " doesn't have 6 columns in row:" + std::to_string(row));
      chr_all.push_back(std::atoi(t[0].c_str()));
      pos_all.push_back(std::atof(t[3].c_str()));
      row++;
    }
  }
  int n_ind = 0;
  {
    std::ifstream in(fam);
    if (!in) die("Problems opening file: " + fam + "\t right directory?");
    std::string line;
    while (std::getline(in, line))
      if (line.find_first_not_of(" \t\r") != std::string::npos) n_ind++;
  }
  const int nsnp = (int)chr_all.size();
  std::ifstream in(bed, std::ios::binary);
  if (!in) die("Couln't open input file: " + bed);
  unsigned char magic[3];
  in.read((char *)magic, 3);
  if (!in || magic[0] != 0x6C || magic[1] != 0x1B) die("Input file does not appear to be a .bed file");
  const bool snp_major = magic[2] != 0;
  const size_t major = snp_major ? nsnp : n_ind, minor = snp_major ? n_ind : nsnp;
  const size_t bytes_per = (minor + 3) / 4;
  std::vector<unsigned char> body(major * bytes_per);
  in.read((char *)body.data(), (std::streamsize)body.size());
  if ((size_t)in.gcount() != body.size()) die("Unexpected end of file reached");
  static const uint8_t recode[4] = {3, 0, 2, 1};  // {1,0,2,3} then 1 <-> 3
  std::vector<int> keep;
  for (int s = 0; s < nsnp; s++)
    if (chr_all[s] != 0 && chr_all[s] < 23) keep.push_back(s);
  d->N = n_ind;
  d->M = (int)keep.size();
  d->pos.resize(d->M);
  d->chr.resize(d->M);
  std::printf("\t-> Plink file contains %d autosomale SNP's\n", d->M);
  if (snp_major) {  // the payload goes to the GPU as it is (relate_hip_load_bed); only the short vectors are handled here
    for (int c = 0; c < d->M; c++) {
      d->pos[c] = pos_all[keep[c]] / 1000000;  // PLINK_POS_SCALING, src/conf.h:8
      d->chr[c] = chr_all[keep[c]];
    }
    d->cols = keep;
    d->bed_rows = nsnp;
    d->bed.assign(body.begin(), body.end());
    return;
  }
  d->geno.assign((size_t)d->N * d->M, 0);
  for (int c = 0; c < d->M; c++) {
    int s = keep[c];
    d->pos[c] = pos_all[s] / 1000000;  // PLINK_POS_SCALING, src/conf.h:8
    d->chr[c] = chr_all[s];
    for (int i = 0; i < n_ind; i++) {
      size_t a = snp_major ? (size_t)s : (size_t)i, b = snp_major ? (size_t)i : (size_t)s;
      unsigned code = (body[a * bytes_per + b / 4] >> (2 * (b % 4))) & 3u;
      d->geno[(size_t)i * d->M + c] = recode[code];
    }
  }
}

// mysort (src/asort.cpp:37-88): order by (chr, pos); equal neighbouring positions abort
void sort_snps(DataSet *d) {
  std::vector<int> id(d->M);
  for (int i = 0; i < d->M; i++) id[i] = i;
  std::sort(id.begin(), id.end(), [&](int a, int b) {
    if (d->chr[a] == d->chr[b]) return d->pos[a] < d->pos[b];
    return d->chr[a] < d->chr[b];
  });
  for (int i = 0; i + 1 < d->M; i++)
    if (d->pos[id[i]] == d->pos[id[i + 1]]) {
      std::printf("\n\t-> Non unique position: chromosome:%d position:%f\n \t   Will exit...\n", d->chr[id[i]],
                  d->pos[id[i + 1]]);
      std::exit(1);
    }
  DataSet s;
  s.N = d->N; s.M = d->M;
  s.geno.resize(d->geno.size());
  s.pos.resize(d->M);
  s.chr.resize(d->M);
  const bool packed = !d->bed.empty();
  if (packed) s.cols.resize(d->M);
  for (int c = 0; c < d->M; c++) {
    s.pos[c] = d->pos[id[c]];
    s.chr[c] = d->chr[id[c]];
    if (packed) s.cols[c] = d->cols[id[c]];  // the sort only permutes the row list of the packed payload
    else
      for (int i = 0; i < d->N; i++) s.geno[(size_t)i * d->M + c] = d->geno[(size_t)i * d->M + id[c]];
  }
  if (packed) {
    s.bed.swap(d->bed);
    s.bed_rows = d->bed_rows;
  }
  *d = std::move(s);
}

// getTestIndividuals (:447-474): all i < j among the flagged individuals, in file order
std::vector<int32_t> keep_list_pairs(const std::string &path) {
  std::vector<std::string> tok;
  read_tokens(path, &tok, nullptr);
  std::vector<int> idx;
  for (size_t i = 0; i < tok.size(); i++)
    if (token_to_int(tok[i]) == 1) idx.push_back((int)i);
  std::vector<int32_t> pairs;
  for (size_t a = 0; a < idx.size(); a++)
    for (size_t b = a + 1; b < idx.size(); b++) { pairs.push_back(idx[a]); pairs.push_back(idx[b]); }
  return pairs;
}

// ostream << setprecision(p) << double  ==  printf("%.{p}g")
void put_g(FILE *f, double v, int prec) { std::fprintf(f, "%.*g", prec, v); }

// write_dMatrix_to_file (src/relateHMM.cpp:117-136) for a decoded block: per pair the IBD0 row and, unless k2 is
// fixed at 0, the IBD1 row (:3771-3782); tab after every value, -1 for SNPs dropped for the pair, 4 significant digits
// (POSTPRECISION).  `post` holds exactly those rows in that order (relate_hip_decode_rows: nb x rpp x M).  The library
// formats the rows on the host threads (printf's bytes, relate_format_rows); groups of pairs bound the text buffer to
// ~256 MiB.
void write_post_block(FILE *f, const double *post, int nb, int M, bool k1_only, std::vector<char> &text,
                      std::vector<size_t> &len) {
  const size_t stride = relate_post_text_stride(M);
  const int rpp = k1_only ? 1 : 2;
  const int group = (int)std::max<size_t>(1, std::min<size_t>((size_t)nb, ((size_t)256 << 20) / (stride * rpp)));
  text.resize((size_t)group * rpp * stride);
  len.resize((size_t)group * rpp);
  for (int g0 = 0; g0 < nb; g0 += group) {
    const int ng = std::min(group, nb - g0);
    if (relate_format_rows(post + (size_t)g0 * rpp * M, (long)ng * rpp, M, text.data(), len.data()))
      die(relate_hip_last_error(nullptr));
    for (int r = 0; r < ng * rpp; r++)
      if (std::fwrite(text.data() + (size_t)r * stride, 1, len[r], f) != len[r]) die("cannot write the .post file");
  }
}

void dump_stripped(relate_hip_t h, const DataSet &d, const std::vector<int> &keep, int M, const std::string &fg,
                   const std::string &fp, const std::string &fc, const std::string &fk) {
  std::vector<uint8_t> g((size_t)d.N * M);
  std::vector<double> pos(M);
  if (relate_hip_get_stripped(h, g.data(), pos.data(), nullptr)) die(relate_hip_last_error(h));
  FILE *f = std::fopen(fg.c_str(), "w");  // iMatrix::print, src/types.h:34-56
  if (!f) die("cannot write " + fg);
  std::vector<char> line((size_t)2 * M + 1);  // "%d " of a value 0..3 is two characters
  for (int i = 0; i < d.N; i++) {
    const uint8_t *row = &g[(size_t)i * M];
    for (int s = 0; s < M; s++) { line[2 * (size_t)s] = (char)('0' + row[s]); line[2 * (size_t)s + 1] = ' '; }
    line[(size_t)2 * M] = '\n';
    if (std::fwrite(line.data(), 1, line.size(), f) != line.size()) die("cannot write " + fg);
  }
  std::fclose(f);
  f = std::fopen(fp.c_str(), "w");  // dArray::print, :160-179
  if (!f) die("cannot write " + fp);
  for (int s = 0; s < M; s++) std::fprintf(f, "%f ", pos[s]);
  std::fputc('\n', f);
  std::fclose(f);
  f = std::fopen(fc.c_str(), "w");
  if (!f) die("cannot write " + fc);
  for (int s = 0; s < d.M; s++)
    if (keep[s]) std::fprintf(f, "%d ", d.chr[s]);
  std::fputc('\n', f);
  std::fclose(f);
  f = std::fopen(fk.c_str(), "w");  // bArray::print
  if (!f) die("cannot write " + fk);
  for (int s = 0; s < d.M; s++) std::fprintf(f, "%d ", keep[s]);
  std::fputc('\n', f);
  std::fclose(f);
}

void usage() {
  std::printf(
      "\tSYNTAX: flatfiles\n\t./relateHMM_hip -o OPTION -g GENOTYPES -p POSITIONS -c CHROMO -d TESTINDIVIDUALS -v VERBOSE\n"
      "\t-post POSTOUTPUT -k KOUTPUT | -file PREFIX | -out PREFIX\n"
      "\tSYNTAX: plink binary files\n\t./relateHMM_hip -o OPTION -plink PREFIX (or -plink-bed/-plink-bim/-plink-fam)\n"
      "\textra: -seed N (pin the start points), -gpu D (device ordinal), -est FILE (a, k2, k1, -logL, runs per pair),\n"
      "\t       -post-block N (pairs per posterior decode call)\n"
      "\tmulti-GPU: one process per GPU, -rank R -nranks N (default: RANK / WORLD_SIZE of the environment); every rank\n"
      "\t       takes a contiguous shard of the pair list and writes its posterior rows to POSTOUTPUT.rankR; the\n"
      "\t       estimates are gathered (-gather rccl | file, rendezvous files next to -rdv PATH) and rank 0 writes\n"
      "\t       KOUTPUT and the -est file for all pairs; -gather none: no exchange, every rank writes KOUTPUT.rankR\n"
      "\t       and the -est file's .rankR for its own shard\n");
}

std::string strip_ext(const std::string &p, std::initializer_list<const char *> exts) {
  for (const char *e : exts) {
    size_t n = std::strlen(e);
    if (p.size() > n && p.compare(p.size() - n, n, e) == 0) return p.substr(0, p.size() - n);
  }
  return p;
}

}  // namespace

int main(int argc, char **argv) {
  for (int i = 0; i < argc; i++) std::printf("%s ", argv[i]);
  std::printf("\n");
  if (argc == 1) { usage(); return 0; }
  std::string geno, pos, chr, option = "options.txt", postF = "postoutput.post", kF = "koutput.k", estF;
  std::string sg = "stripped.geno", sp = "stripped.pos", sc = "stripped.chr", sk = "keep.index";  // src/conf.h:16-19
  std::string bed, bim, fam, keepfile;
  int plink = 0, verbose = 1, device = 0;
  long post_block = 0;  // -post-block: pairs per decode call (default: what ~1 GiB of posterior rows holds)
  bool have_seed = false;
  unsigned seed = 0;
  auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e && *e ? std::atoi(e) : dflt; };
  int rank = env_int("RANK", 0), nranks = env_int("WORLD_SIZE", 1);
  bool have_gpu = false;
  std::string gatherT = "rccl", rdv;
  for (int a = 1; a < argc; a += 2) {
    std::string f = argv[a];
    if (f == "--help") { usage(); return 0; }
    if (a + 1 >= argc) { std::printf("\nArgument without value will exit: %s \n", argv[a]); return 1; }
    std::string v = argv[a + 1];
    if (f == "-o") option = v;
    else if (f == "-g") geno = v;
    else if (f == "-p") pos = v;
    else if (f == "-c") chr = v;
    else if (f == "-post") postF = v;
    else if (f == "-k") kF = v;
    else if (f == "-v") verbose = std::atoi(v.c_str());
    else if (f == "-plink-bed") { bed = v; plink++; }
    else if (f == "-plink-bim") { bim = v; plink++; }
    else if (f == "-plink-fam") { fam = v; plink++; }
    else if (f == "-out") {
      postF = v + ".post"; kF = v + ".k";
      sg = v + "_stripped.geno"; sc = v + "_stripped.chr"; sp = v + "_stripped.pos"; sk = v + "_stripped.keep";
    } else if (f == "-file") {
      std::string front = strip_ext(v, {".geno", ".chr", ".pos"});
      pos = front + ".pos"; chr = front + ".chr"; geno = front + ".geno";
    } else if (f == "-plink") {
      std::string front = strip_ext(v, {".bed", ".bim", ".fam"});
      bim = front + ".bim"; fam = front + ".fam"; bed = front + ".bed";
      plink += 3;
    } else if (f == "-choose") { /* both routes are the fast one here */ }
    else if (f == "-d") keepfile = v;
    else if (f == "-seed") { seed = (unsigned)std::strtoul(v.c_str(), nullptr, 10); have_seed = true; }
    else if (f == "-gpu") { device = std::atoi(v.c_str()); have_gpu = true; }
    else if (f == "-rank") rank = std::atoi(v.c_str());
    else if (f == "-nranks") nranks = std::atoi(v.c_str());
    else if (f == "-gather") gatherT = v;
    else if (f == "-rdv") rdv = v;
    else if (f == "-est") estF = v;
    else if (f == "-post-block") post_block = std::atol(v.c_str());
    else { std::printf("\nArgument unknown will exit: %s \n", argv[a]); usage(); return 1; }
  }
  if (nranks < 1 || rank < 0 || rank >= nranks) die("-rank must lie in 0 .. -nranks - 1");
  if (gatherT != "rccl" && gatherT != "file" && gatherT != "none") die("-gather takes rccl, file or none");
  // -gather none: no exchange at all -- every rank writes the estimates of its own shard to KOUTPUT.rank<r> and the
  // -est file's .rank<r> (ranks started apart, e.g. as array jobs; laid end to end the shards are the whole list)
  const bool own_out = gatherT == "none" && nranks > 1;
  if (!have_gpu && nranks > 1) device = env_int("LOCAL_RANK", rank);  // one process per GPU
  if (verbose) {
    time_t t = time(nullptr);
    std::printf("\t-> relateHMM_hip (C ABI over gfx950 kernels; reference CLI 0.992 semantics)\t%s", ctime(&t));
    if (nranks > 1) std::printf("\t-> rank %d of %d on GPU %d\n", rank, nranks, device);
  }
  DataSet d;
  const bool chr_given = !chr.empty();
  if (!plink) {
    if (geno.empty()) die("You must supply atleast a genotype file -g , will exit");
    if (pos.empty()) die("You must supply atleast a positionfile file -p , will exit");
    std::vector<std::string> tok;
    read_tokens(pos, &tok, nullptr);
    for (auto &t : tok) d.pos.push_back(token_to_double(t));
    if (chr_given) {
      tok.clear();
      read_tokens(chr, &tok, nullptr);
      for (auto &t : tok) d.chr.push_back(token_to_int(t));
    } else {
      if (verbose) std::printf("\t-> no chromosomefile given, will assume all loci is on same chromosome\n");
      d.chr.assign(d.pos.size(), -1);
    }
    read_geno_text(geno, &d);
  } else {
    if (!geno.empty() || !pos.empty() || chr_given)
      die("You must use either plink binary file or -g,-p file as input, will exit");
    if (plink != 3) die("you need to supply all .bim .bed .fam files");
    read_plink(bed, bim, fam, &d);
    sort_snps(&d);
  }
  if (d.M != (int)d.chr.size() || d.M != (int)d.pos.size())
    die("Dimension of data input doesn't have compatible dimensions, program will exit\nDimension of genodata:=(" +
        std::to_string(d.N) + "," + std::to_string(d.M) + "), positions:=" + std::to_string(d.pos.size()) +
        ", chromosomes:=" + std::to_string(d.chr.size()));
  Options op;
  set_defaults(&op);
  read_options(option, &op);
  if (chr_given && !plink) sort_snps(&d);
  op.o.seed_mode = 0;
  op.o.seed = have_seed ? seed : (unsigned)time(nullptr);
  if (verbose)
    std::printf("\t-> Dimension of genodata:=(%d,%d), positions:=%d, chromosomes:=%d \n", d.N, d.M, (int)d.pos.size(),
                (int)d.chr.size());
  if (op.pair[0] < 0 || op.pair[0] >= d.N || op.pair[1] < 0 || op.pair[1] >= d.N)
    die("You want to run analysis on pair=(" + std::to_string(op.pair[0]) + "," + std::to_string(op.pair[1]) +
        "), genotype dataset only contains (0..." + std::to_string(d.N) + "), will exit");

  relate_hip_t h = nullptr;
  if (relate_hip_create(&h, device)) die(relate_hip_last_error(nullptr));
  std::vector<int> keep(d.M);
  int M = 0;
  int lrc = d.bed.empty()
                ? relate_hip_load(h, d.geno.data(), d.N, d.M, d.pos.data(), d.chr.data(), &op.o)
                : relate_hip_load_bed(h, d.bed.data(), d.N, d.bed_rows, d.cols.data(), d.M, d.pos.data(), d.chr.data(),
                                      &op.o);
  if (lrc || relate_hip_precalc(h, keep.data(), &M))
    die(relate_hip_last_error(h));

  std::vector<int32_t> pairs;
  const bool batch = op.doAllPairs || !keepfile.empty();
  if (!keepfile.empty()) {
    pairs = keep_list_pairs(keepfile);
    for (int32_t v : pairs)
      if (v >= d.N) die("keep list names individual " + std::to_string(v) + " but the data has " + std::to_string(d.N));
  } else if (op.doAllPairs) {
    for (int i = 0; i < d.N - 1; i++)
      for (int j = i + 1; j < d.N; j++) { pairs.push_back(i); pairs.push_back(j); }
  } else {
    pairs = {op.pair[0], op.pair[1]};
  }
  // Multi-GPU (no counterpart in the reference, whose pair loop is one process: src/relateHMM.cpp:3764-3785,
  // 3851-3875): rank r takes the contiguous shard relate_hip_shard gives it.  Every rank loads the data set and
  // runs the pre-calculation itself (inputs are replicated, nothing is exchanged until the estimates are gathered).
  const long Pall = (long)pairs.size() / 2;
  int64_t sb = 0, se = Pall;
  relate_hip_shard(Pall, rank, nranks, &sb, &se);
  const std::vector<int32_t> all_pairs = pairs;
  pairs.assign(all_pairs.begin() + 2 * sb, all_pairs.begin() + 2 * se);
  const long P = (long)pairs.size() / 2;
  if (nranks > 1 && !batch) die("-nranks > 1 needs a pair list (all pairs or -d)");
  if (rdv.empty()) rdv = kF + ".rdv";
  if (batch && rank > 0) {  // the dumps and the names of the outputs are rank 0's business
    if (!postF.empty()) postF = postF + ".rank" + std::to_string(rank);  // (-post "": no posterior rows at all)
    std::printf("\t-> rank %d: pairs %ld .. %ld of %ld\n", rank, (long)sb, (long)se, Pall);
  } else if (batch) {
    std::printf("\t-> Will dump these data, that are used for analysis after the maf/prune removal\n");
    if (!keepfile.empty()) { sg = "stripped.geno"; sp = "stripped.pos"; sc = "stripped.chr"; sk = "keep.index"; }
    std::printf("\t\tgeno:\t\"%s\"\n\t\tchromo:\t\"%s\"\n\t\tpos:\t\"%s\"\n\t\tkeepList:\"%s\"\n", sg.c_str(), sc.c_str(),
                sp.c_str(), sk.c_str());
    // (joblist_fast_all_pairs dumps to the default names whatever -out says, src/relateHMM.cpp:3832-3839)
    dump_stripped(h, d, keep, M, sg, sp, sc, sk);
    // (with several ranks the row shards keep the name given, so that rank r's file is POSTOUTPUT.rank<r> for all r)
    if (!postF.empty()) postF = nranks > 1 ? postF + ".rank0" : fresh_name(postF);
    kF = fresh_name(kF);
    std::printf("\t-> number of individuals:%d\n\t-> number of pairs to run:%ld\n", d.N, Pall);
  } else {
    if (file_exists(kF)) std::printf("\t-> Filename: %s exists will append new results to file\n", kF.c_str());
    if (file_exists(postF)) std::printf("\t-> Filename: %s exists will append new results to file\n", postF.c_str());
  }
  // A rank's row shard starts empty: it must hold this run's rows and nothing else (rank r's rows are found by
  // position in POSTOUTPUT.rank<r>).  The single-process file is appended to, under a name fresh_name made unused.
  FILE *fpost = postF.empty() ? nullptr : std::fopen(postF.c_str(), nranks > 1 ? "w" : "a");
  if (own_out) {
    kF += ".rank" + std::to_string(rank);
    if (!estF.empty()) estF += ".rank" + std::to_string(rank);
  } else if (rank > 0) {
    kF.clear();
    estF.clear();
  }
  FILE *fk = kF.empty() ? nullptr : std::fopen(kF.c_str(), own_out ? "w" : "a");
  FILE *fest = estF.empty() ? nullptr : std::fopen(estF.c_str(), "w");
  if ((!postF.empty() && !fpost) || (!kF.empty() && !fk) || (!estF.empty() && !fest)) die("cannot open output files");
  const bool k1_only = op.o.fixK == 1 && op.o.fixK_val == 0;  // :3771
  // The fit takes the pairs in large blocks (its launches need tens of thousands of pairs in flight to fill the
  // GPU); the posteriors of a fitted block are decoded in sub-blocks that bound the host buffer to ~1 GiB.
  const int rpp = k1_only ? 1 : 2;  // posterior rows the .post file takes per pair: only those are decoded and fetched
  const long FB = std::max<long>(1, std::min<long>(P, 65536));
  long B = std::max<long>(1, std::min<long>(FB, ((long)1 << 30) / ((long)rpp * M * (long)sizeof(double))));
  if (post_block > 0) B = std::min<long>(FB, post_block);
  std::vector<relate_fit> mine((size_t)std::max<long>(P, 1));  // this rank's records, in pair order
  std::vector<double> par((size_t)B * 3);
  double *post = nullptr;  // page-locked: the posterior rows arrive by direct DMA
  if (fpost && relate_hip_host_alloc((void **)&post, (size_t)B * rpp * M * sizeof(double)))
    die(relate_hip_last_error(nullptr));
  std::vector<int> ns((size_t)B);
  std::vector<char> text;           // formatted rows of a group of pairs (write_post_block)
  std::vector<size_t> text_len;
  int bad = 0, nosnp = 0;
  for (long f0 = 0; f0 < P; f0 += FB) {
    const long nf = std::min<long>(FB, P - f0);
    if (relate_hip_fit(h, &pairs[2 * f0], (int)nf, nullptr, &mine[(size_t)f0], nullptr)) die(relate_hip_last_error(h));
    for (long b0 = f0; b0 < f0 + nf; b0 += B) {
      int nb = (int)std::min<long>(B, f0 + nf - b0);
      relate_fit *fit = &mine[(size_t)b0];
      for (int p = 0; p < nb; p++) { par[3 * p] = fit[p].a; par[3 * p + 1] = fit[p].k2; par[3 * p + 2] = fit[p].k1; }
      if (fpost) {  // the posterior rows (the SNP count of the .est line comes with the fit)
        if (relate_hip_decode_rows(h, &pairs[2 * b0], nb, par.data(), rpp, post, ns.data()))
          die(relate_hip_last_error(h));
      }
      if (fpost) write_post_block(fpost, post, nb, M, k1_only, text, text_len);
      if (verbose) {
        std::printf("\r\t-> %ld/%ld pairs ", b0 + nb, P);
        std::fflush(stdout);
      }
    }
  }
  if (verbose) std::printf("\n");
  // the estimates of all ranks, in pair order, on every rank (the path's one exchange); rank 0 writes them
  std::vector<relate_fit> gathered;
  const relate_fit *allfit = mine.data();
  if (nranks > 1 && !own_out) {
    gathered.resize((size_t)std::max<long>(Pall, 1));
    if (relate_hip_gather(h, mine.data(), Pall, rank, nranks, rdv.c_str(), gatherT == "file" ? 1 : 0, gathered.data()))
      die(relate_hip_last_error(h));
    allfit = gathered.data();
  }
  // what this process writes: all pairs on rank 0 after the gather, its own shard under -gather none
  const long w0 = own_out ? (long)sb : 0, w1 = own_out ? (long)se : Pall;
  if (rank == 0 || own_out)
    for (long p = w0; p < w1; p++) {
      const relate_fit &f = own_out ? mine[(size_t)(p - sb)] : allfit[p];
      if (f.status == RELATE_E_NOSNP) nosnp++;
      else if (f.status) bad++;
      if (fk) {
        // batch drivers: setprecision(POSTPRECISION) (src/relateHMM.cpp:142); lone pair: default precision
        // (src/runRelate.cpp:70)
        int prec = batch ? 4 : 6;
        put_g(fk, f.k1, prec); std::fputc('\n', fk);
        if (!k1_only) { put_g(fk, f.k2, prec); std::fputc('\n', fk); }
      }
      if (fest)
        std::fprintf(fest, "%d\t%d\t%.17g\t%.17g\t%.17g\t%.17g\t%d\t%d\t%d\t%d\n", all_pairs[2 * p], all_pairs[2 * p + 1],
                     f.a, f.k2, f.k1, f.negloglik, f.timesRun, f.timesConverged, f.nsnp, f.status);
    }
  if (bad) std::printf("\t-> %d pair(s) hit a NaN in the recursion (the reference would have exited there)\n", bad);
  if (nosnp)  // src/relateHMM.cpp:3415-3421: the reference stops at the first such pair
    std::printf("\t-> %d pair(s) don't share 1 non-missing SNP: not fitted, rows of -1 (the reference would have exited there)\n", nosnp);
  if (fpost) std::fclose(fpost);
  if (fk) std::fclose(fk);
  if (fest) std::fclose(fest);
  relate_hip_host_free(post);
  relate_hip_destroy(h);
  if (verbose) {
    time_t t = time(nullptr);
    std::printf("\t-> %s", ctime(&t));
  }
  return 0;
}
