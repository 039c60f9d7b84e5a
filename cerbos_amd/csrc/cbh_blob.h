// Layout of the lowered policy table image ("blob") shared by the lowering step
// (cerbos_amd/lower/blob.py writes it) and the device code (cbh_engine.hip reads it).
// Little-endian, every section 64-byte aligned.  The image is position independent: it is
// copied to HBM verbatim (or broadcast GPU->GPU) and addressed as base + section offset.
#pragma once
#include <stdint.h>

#define CBH_BLOB_MAGIC 0x31484243u /* "CBH1" */
#define CBH_BLOB_VERSION 22u

struct CbhBlobHeader {  // 32 bytes
  uint32_t magic;
  uint32_t version;
  uint32_t n_sections;
  uint32_t reserved0;
  uint64_t total_len;
  uint64_t reserved1;
};

struct CbhBlobSection {  // 32 bytes
  uint32_t id;
  uint32_t count;  // element count (section specific)
  uint64_t offset; // from blob start
  uint64_t nbytes;
  uint64_t reserved;
};

enum CbhSectionId {
  CBH_SEC_META = 1,        // u32[CBH_META_N]
  CBH_SEC_STR_OFF = 2,     // u32[K+1]
  CBH_SEC_STR_BYTES = 3,   // u8[]
  CBH_SEC_SCOPE_PARENT = 4, // u32[NS]  parent scope index or CBH_NONE
  CBH_SEC_SCOPE_FLAGS = 5, // u32[NS]  bit0 resource map, bit1 principal map, bits 2..3 scope permissions
  CBH_SEC_SCOPE_SID = 6,   // u32[NS]  string id of the scope
  CBH_SEC_HASH = 7,        // CbhHashSlot[nslots]
  CBH_SEC_ROWS = 8,        // u32[n_rows][16]  row-major records (CbhRowField order): two s_load_dwordx8 halves
  CBH_SEC_RPROWS = 9,      // u32[n_rprows][4] row-major records (CbhRpField order)
  CBH_SEC_U32POOL = 10,    // u32[] (pattern lists, parent-role lists)
  CBH_SEC_DR = 11,         // u32[n_dr][4]     row-major records (CbhDrField order)
  CBH_SEC_CODE = 12,       // u32[]
  CBH_SEC_CONST_TAG = 13,  // u8[]
  CBH_SEC_CONST_VAL = 14,  // u64[]
  CBH_SEC_THEAP_TAG = 15,  // u8[]
  CBH_SEC_THEAP_VAL = 16,  // u64[]
  CBH_SEC_GBITS = 17,      // u64[3][K]  glob match bits of table strings per dimension
  CBH_SEC_NFA_ACTION = 18, // see CbhNfa layout below
  CBH_SEC_NFA_ROLE = 19,
  CBH_SEC_NFA_KIND = 20,
  CBH_SEC_POLICY_SID = 21, // u32[n_policies] string ids of policy keys (host decode)
  CBH_SEC_DRNAME_SID = 22, // u32[n_drnames]  string ids of derived role names (bit order of edr_mask)
  CBH_SEC_CONST_REC = 23,  // u32[n_consts][4]  {tag, 0, lo, hi}: the constant pool as scalar-loadable records
  CBH_SEC_THEAP_REC = 24,  // u32[theap_len][4] the constant heap, same record form
  CBH_SEC_ROLE_CLASS = 25, // u8[K] class (0..61) of a string that is a literal rule role, 63 = any other string
  CBH_SEC_ROWLEAF2 = 30,     // u32[n_rows][8]  copy of the fused-leaf record of a rule's DERIVED-ROLE condition (CBH_ROW_F_DRLEAF_EMBEDDED)
  CBH_SEC_DRX = 31,          // u32[n_dr][16]   derived-role definitions for the flat kernel (CbhDrxField order)
  CBH_SEC_REGEX = 33,        // u32[]           DFA tables of constant `matches` patterns, back to back: {n_states, n_classes,
                             //                 classmap (256 bytes in 64 words), flags[n_states], trans[n_states][n_classes]}
  // Trace pass (cbh_trace_batch: which CEL errors were absorbed, which rule outputs fired).  Its programs live on the same
  // tape but keep what the decision programs fuse away: every leaf carries the id of its expression text (OP_LEAF arg =
  // trace string id + 1), an inlined variable is closed by OP_VARSCOPE, an output expression by OP_OUT.
  CBH_SEC_TRACE_ROWS = 34,   // u32[n_rows][8]   {cond, drcond, vars_off, vars_cnt, drvars_off, drvars_cnt, out_activated, out_not_met}:
                             //                  program entries (CBH_NONE = none) and slices of CBH_SEC_TRACE_POOL
  CBH_SEC_TRACE_DR = 35,     // u32[n_dr][4]     {cond, vars_off, vars_cnt, 0}
  CBH_SEC_TRACE_RP = 36,     // u32[n_rprows][8] {cond, vars_off, vars_cnt, out_activated, out_not_met, 0, 0, 0}
  CBH_SEC_TRACE_POOL = 37,   // u32[]            entries of the variable programs of a params set, in definition order
  CBH_SEC_TRACE_STRINGS = 38, // host only, JSON {"strings": [...], "templates": {"<rule word>": [template, parts]}}: the expression texts,
                              // variable names and rule FQNs the trace records refer to, and how the consumer assembles an output
                              // value from its parts (template = ["hole", j] | ["const", v] | ["list", [t..]] | ["map", [[kt, vt]..]] |
                              // ["format", "<fmt>", [t..]])
  CBH_SEC_TRACE_HOST = 39,    // host only (cbh_ingest.cpp cbi_trace_pb): the same strings and templates in binary - u32 n, {u32 len, bytes}*;
                              // u32 n_templates, {u32 rule word, u32 parts, node}*; node = u8 kind: 0 hole + u32 j | 1 const + u8 type (0 null,
                              // 1 bool + u8, 2 int + i64, 3 double + f64, 4 string + u32 len + bytes) | 2 list + u32 n + nodes | 3 map + u32 n +
                              // (key node, value node)* | 4 format + u32 len + bytes + u32 n + nodes
  CBH_SEC_ROWX = 40,         // u32[n_rows][8]  what cbh_check_walk2.h reads besides the record (CbhRowXField); rows with CBH_ROW_F_X
  CBH_SEC_RPX = 41,          // u32[n_rprows][16] role-policy rules for that kernel (CbhRpxField)
  CBH_SEC_STR_WFLAGS = 42,   // u8[K] CBH_SWF_*: what the walk would otherwise probe the directory for, lane by lane
  // A FLAT table's buckets once more as SEGMENTS of up to 64 consecutive records each, held the way the reference's own index
  // holds its rows - one bitmap per dimension value (index/index.go:270-305) - for the flat kernel's mask walk
  // (cbh_check_flat.h).  Bit i of every mask = record i of the segment, in binding order.  A segment is one block, a multiple
  // of 16 dwords: {CbhSegHdr}{u64 action_mask[32]}{u64 role_mask[32]}{u8 item of record i's condition [64]}{u8 ... of its
  // derived-role condition [64]} (0xFF = none, or an item the lanes do not decide by themselves) {CbhSegDesc[n_items]}
  // {u32 ref[n_items]: the items' condition references} {CbhSegItem[n_complex], on a 16-dword boundary}{CbhLeaf4[n_leaves],
  // padded to a multiple of four}; the segments of a bucket follow each other (CBH_B_RESSEG: first block, count).
  CBH_SEC_SEGS = 43,
  CBH_SEC_LEAFPOOL = 44,     // CbhLeaf4[n] the distinct fused leaves of a POOLED table's conditions (CBH_M_SEGS), n <= 64, padded to a multiple of four
  CBH_SEC_ROWPAT = 29,       // u32[n_rows][8]  pattern halves of the rule records (CbhRowPatField order)
  CBH_SEC_ACTION_CLASS = 28, // u8[K] class (0..61) of a string that is a literal rule action of a resource policy, 63 = any other string
  CBH_SEC_HOST_NAMES = 27,   // host only: {u32 n, {u16 len, bytes}*} policy keys (CBH_P_TABLE ids), then the same for derived-role names
  CBH_SEC_COLUMN_PATHS = 26, // host only (cbh_ingest.cpp): per column {u8 root 0=P.attr 1=R.attr 2=auxData.jwt 3=auxData.jwts, u8 n_keys, {u16 len, bytes}*}
};

enum CbhMeta {
  CBH_M_NSTRINGS = 0,
  CBH_M_NCOLUMNS = 1,
  CBH_M_NSCOPES = 2,
  CBH_M_HASH_MASK = 3, // nslots - 1 (power of two)
  CBH_M_NROWS = 4,
  CBH_M_NRPROWS = 5,
  CBH_M_NDR = 6,
  CBH_M_NPOLICIES = 7,
  CBH_M_NCONSTS = 8,
  CBH_M_CODE_LEN = 9,
  CBH_M_FLAGS = 10,     // bit0: some program reads runtime.effectiveDerivedRoles
  CBH_M_MAX_STACK = 11,
  CBH_M_NDRNAMES = 12,
  CBH_M_NFA_WORDS_ACTION = 13, // u64 words of NFA state per dimension (0 = no globs)
  CBH_M_NFA_WORDS_ROLE = 14,
  CBH_M_NFA_WORDS_KIND = 15,
  CBH_M_THEAP_LEN = 16,
  CBH_M_MAX_LOCALS = 17,
  CBH_M_GSLOTS_GENERIC = 18, // evaluation-site slots (cbh_check_walk2.h) of sites the walk cannot decide inline for ANY batch: slots 0 .. n - 1
  CBH_M_GSLOTS_ALL = 19,     // ... plus the sites it can decide inline for plain scalars only: slots 0 .. n - 1
  CBH_M_INLINE_COLS = 20,    // the inline leaf code of the flat / walk2 kernels reads attribute columns 0 .. n - 1 only (the lowering
                             // numbers those first): what a walk without generic programs parks in LDS
  CBH_M_Q_SITES = 22,        // CBH_BS_ROW_GENERIC / _OPEN: the role-policy rules hold evaluation sites of that kind; CBH_BS_DR_GENERIC /
                             // _OPEN (bits 3, 4): the principal policies' rules do
  CBH_M_SENS_COLS = 21,      // bit c: an int / uint / list / map value in column c sends a classified leaf to the shared evaluator -
                             // the columns the host looks at to call a batch "plain" (cbh_engine.hip validate_batch)
  CBH_M_SEGS = 23,          // CBH_MSEG_PRESENT: the table has CBH_SEC_SEGS; CBH_MSEG_POOLED: its conditions are built from at most 64
                            // distinct fused leaves - numbered table-wide, in CBH_SEC_LEAFPOOL (bits 0..7: how many), the segments
                            // carry none
  CBH_META_N = 24
};
#define CBH_MSEG_PRESENT 0x80000000u
#define CBH_MSEG_POOLED 0x40000000u
#define CBH_SEG_RECORDS 64u
#define CBH_SEG_TAIL_PAD16 20u  /* 16-dword units of zeros behind the last block of CBH_SEC_SEGS: a lane loads its descriptor slot whatever n_items */
#define CBH_SEG_COMPLEX 16u     /* CbhSegItem entries of one segment at most */
#define CBH_SEG_FIXED_DWORDS 176u   /* header + class masks + the two record -> item tables; the item descriptors follow */
struct CbhSegHdr {   // 16 dwords
  uint32_t allow_lo, allow_hi;       // the segment's ALLOW records
  uint32_t deny_lo, deny_hi;         // ... its DENY records
  uint32_t simple_c_lo, simple_c_hi; // the records whose condition is an item a lane decides by itself (CbhSegDesc)
  uint32_t simple_d_lo, simple_d_hi; // ... whose derived-role condition is
  uint32_t n_items, n_leaves;        // distinct conditions / fused leaves of the segment (n_leaves = 0 in a pooled table)
  uint32_t size16;                   // the block's size in 16-dword units: the next segment of the bucket starts there
  uint32_t n_records, row_begin;     // the records CBH_SEC_ROWS[row_begin .. row_begin + n_records)
  uint32_t n_complex;                // CbhSegItem entries: the conditions the wave evaluates as one (deeper trees, more than four leaves, programs)
  uint32_t off_refs_complex;         // dword offsets from the block's start: ref[] | CbhSegItem[] << 16
  uint32_t off_leaves;               // ... of the segment's own fused-leaf records
};
struct CbhLeaf4 {    // 4 dwords: a classified fused leaf as the mask walk evaluates it (blob.py _compact_leaves)
  uint32_t w;        // bits 0..3 class (celc.py _leaf_class 1, 2, 3, 4, 6; 0 = none of them), 4..11 the comparison (OP_EQ ..), 16..23 column, 24..31 second column
  uint32_t c[3];     // class 1: constant's tag, low dword; class 2: the double's dwords; class 6: up to three string ids (CBH_NONE beyond)
};
struct CbhSegDesc {  // 2 dwords: ONE level of at most four classified leaves, in order (cbh_check_flat.h lane_items)
  uint8_t leaf[4];   // numbers in the pool, or indices into the segment's leaves
  uint8_t flags;     // bits 0..2 number of leaves (0 = not such an item: a CbhSegItem holds it), bit 3 any instead of all, bit 4 negated
  uint8_t pad[3];
};
struct CbhSegItem {  // 16 dwords: one condition of a segment that the wave evaluates as one
  uint32_t cc_lo, cc_hi;         // the records it is the rule condition of
  uint32_t cd_lo, cd_hi;         // ... the derived-role condition of
  uint32_t how;                  // bits 0..1: 1 = ONE level of classified leaves in leaf_idx order (bit 8: any instead of all, bit 9: the result
                                 // negated - none(..) = not any(..); one fused leaf = all(leaf)), 2 = a deeper tree (ops + leaf_idx), 0 = neither
  uint32_t ref;                  // the condition reference (what the shared evaluator runs where the inline code cannot decide)
  uint32_t id;                   // its index among the segment's items
  uint32_t n_leaves;
  uint32_t leaf_idx[2];          // 8 x u8: the tree's leaves in order - numbers in the pool, or indices into the segment's leaves
  uint32_t pad[2];
  uint32_t ops[4];               // the tree's 4-bit ops (CBH_ROW_F_TREE_EMBEDDED)
};
#define CBH_MF_USES_RUNTIME_EDR 1u
#define CBH_MF_HAS_PARENT_ROLES 2u
#define CBH_MF_HAS_ROLE_POLICIES 4u
#define CBH_MF_HAS_GENERIC_PROGRAMS 8u  /* some program needs the operand-stack interpreter */
#define CBH_MF_HAS_ANY_PATTERN 16u      /* some pattern reference is CBH_PAT_ANY (treated as a glob table) */
#define CBH_MF_HAS_PRINCIPAL_POLICIES 32u
#define CBH_MF_NEEDS_STRING_BYTES 128u   /* glob automata, or a program that looks inside a string: upload str_off / str_bytes / str_flags */
#define CBH_MF_NEEDS_ARENA 1024u          /* some program builds a list (filter, map, intersect, except, list +): the kernels that run the
                                          * operand-stack interpreter get CBH_ARENA_ENTRIES values of LDS per lane for them (cbh_vm.h) */
#define CBH_MF_FLAT_CLOSED 512u           /* FLAT and every condition is evaluated inline by the flat kernel: no evaluator call needed for plain batches */
#define CBH_MF_TRACE_ALL 4096u            /* the table has variables or output expressions: a kernel that cannot tell which inputs need the trace
                                          * pass marks every tuple CBH_ST_WANTS_TRACE (cbh_walk2_kernel can tell) */
#define CBH_MF_WALK2 2048u                /* cbh_check_walk2.h decides this table (batch shape and mode permitting) */
#define CBH_MF_FLAT 256u                  /* resource policies only, leaf conditions, every record decided by class masks: cbh_check_flat.h */
#define CBH_MF_READS_REQUEST_STRINGS 64u /* some program reads a raw request string (CBH_RQ_S_*): upload those fields */

// Directory: open addressing, linear probing, key.x == CBH_NONE marks an empty slot.
struct CbhHashSlot { // 32 bytes
  uint32_t k0, k1, k2, k3; // k0 = bucket type
  uint32_t v0, v1, v2, v3;
};
enum CbhBucketType {
  CBH_B_RESOURCE = 1,  // (ver sid, kind sid, scope idx) -> v0 row_begin, v1 row_count, v2 dr_begin, v3 dr_count
  CBH_B_PRINCIPAL = 2, // (ver sid, scope idx, principal sid) -> v0 row_begin, v1 row_count
  CBH_B_ROLEPOL = 3,   // (ver sid, scope idx, role sid) -> v0 rprow_begin, v1 rprow_count, v2 policy id, v3 U32POOL offset of {probe program
                       // of the policy's variables, its site slot} or CBH_NONE
  CBH_B_PPEXISTS = 4,  // (ver sid, scope idx, 0) -> exists (any principal policy row)
  CBH_B_RPRES = 5,     // (ver sid, scope idx, 0) -> v0 off, v1 cnt into U32POOL of resource pattern refs of role-policy rows
  CBH_B_PARENTS = 6,   // (scope idx, role sid, 0) -> v0 off, v1 cnt into U32POOL of ancestor role sids; v2, v3 = OR of their role classes
  CBH_B_RESEXISTS = 7, // (ver sid, kind sid, scope idx): same key as RESOURCE, present for every resource policy; v1, v2 = union of the
                       // literal role class masks of its rules, v3 = union of their role glob masks (Index.Query's base test);
                       // v0 = 1 | CBH_BS_*: the evaluation sites the bucket holds (cbh_check_walk2.h: the pre-pass skips the rest)
  CBH_B_FAMILY = 9,    // (ver sid, kind sid, 0) -> v0 = OR of CBH_BS_* over the buckets of the family's scopes: a request whose
                       // family holds no site the batch files needs no pre-pass walk; v1, v2 = union of the role classes of the
                       // records / definitions that carry a GENERIC site, v3 bit 0 = that mask can be used (no role glob among them):
                       // a request none of whose role sets meets it cannot reach one
  CBH_B_RPROLES = 8,   // (ver sid, scope idx, 0) -> v0 off, v1 cnt into U32POOL: the roles with a role policy at that scope, sorted by
                       // name (the order of a role's ancestor list, ruletable/build.py)
  CBH_B_RESSEG = 10,   // (ver sid, kind sid, scope idx), a table with CBH_SEC_SEGS: same key as RESOURCE -> v0 the bucket's first segment
                       // block (16-dword units of CBH_SEC_SEGS), v1 segments, v2 dr_begin, v3 dr_count
};

// Condition reference (row / derived-role cond fields): bit31 set -> the program at (ref & ~bit31)
// is a single OP_LEAF_BIN + OP_RET and may be evaluated inline.
#define CBH_COND_LEAF 0x80000000u
#define CBH_COND_LEAFTREE 0x40000000u  /* all/any/none tree of fused leaves: inline, no operand stack */
#define CBH_COND_PC_MASK 0x3FFFFFFFu

// Pattern reference: bit31 set -> glob index within the dimension, else literal string id.
#define CBH_PAT_GLOB 0x80000000u
// The lone "*" (= "**", util/globs_common.go:74-81: matches every string) needs no automaton.
#define CBH_PAT_ANY 0x7FFFFFFFu

// Regular rows (resource + principal policies).  One record stands for the rule-table rows of ONE
// rule that share effect / condition / derived-role condition: the cross product of its role list
// and its action list.  A record is two 8-dword halves:
//   hot half   what every visit needs: effect, condition references, and the role / action lists as
//              64-bit CLASS masks (CBH_SEC_ROLE_CLASS / CBH_SEC_ACTION_CLASS give every literal role /
//              action of the table a class number < 62; 63 = any other string).  When a list is all
//              literals with a class (or "*": every bit set) the mask decides the match exactly
//              (CBH_ROW_F_ROLE_BY_CLASS / _ACTION_BY_CLASS) and the second half is never read;
//   leaf half  a copy of the 8-dword fused-leaf record of the rule's condition when that is a single leaf.
// The pattern references themselves (up to three inline, longer lists in U32POOL) live in a parallel section,
// CBH_SEC_ROWPAT: they are read only for records with glob patterns or more classes than fit, and for
// principal-policy rows.
enum CbhRowField {
  CBH_ROW_FLAGS = 0,    // bits 0..1 effect (1 ALLOW, 2 DENY), CBH_ROW_F_*
  CBH_ROW_COND = 1,     // program entry or CBH_NONE
  CBH_ROW_DRCOND = 2,   // program entry or CBH_NONE
  CBH_ROW_POLICY = 3,   // policy id of the origin policy (strict-mode attribution)
  CBH_ROW_ROLE_CLASSES = 4,    // u64 (2 dwords): role classes the record's role list can match
  CBH_ROW_ACTION_CLASSES = 6,  // u64 (2 dwords): action classes its action list can match
  CBH_ROW_LEAF = 8,     // 8 dwords: a copy of the condition's fused-leaf record (CBH_ROW_F_LEAF_EMBEDDED) - the
                        // visit that needs the condition has it without a second, dependent load
  CBH_ROW_NF = 16       // record = 16 dwords, 64-byte aligned
};
enum CbhRowPatField {   // CBH_SEC_ROWPAT: the pattern half of record i, 8 dwords
  CBH_PAT_ACTION = 0,   // pattern ref (action dim), or U32POOL offset of a list of them (CBH_ROW_F_ACTION_LIST)
  CBH_PAT_ROLE = 1,     // pattern ref (role dim), or U32POOL offset of a list of them (CBH_ROW_F_ROLE_LIST)
  CBH_PAT_RESOURCE = 2, // pattern ref (kind dim) - tested for principal-policy rows only
  CBH_PAT_COUNTS = 3,   // action list length | role list length << 16 (0 = one inline reference)
  CBH_PAT_A1 = 4,       // 2nd, 3rd action pattern ref of a list of at most three (inline, no CBH_ROW_F_ACTION_LIST)
  CBH_PAT_R1 = 6,       // 2nd, 3rd role pattern ref likewise
  CBH_PAT_NF = 8
};
#define CBH_ROW_INLINE_MAX 3u           /* longer lists live in U32POOL */
#define CBH_ROW_F_ACTION_LIST 4u
#define CBH_ROW_F_ROLE_LIST 8u
#define CBH_ROW_F_ROLE_BY_CLASS 16u     /* the role class mask decides the role match exactly */
#define CBH_ROW_F_ACTION_BY_CLASS 32u   /* the action class mask decides the action match exactly */
#define CBH_ROW_F_LEAF_EMBEDDED 64u     /* dwords 8..15 hold the condition's fused-leaf record */
#define CBH_ROW_F_DRLEAF_EMBEDDED 128u  /* CBH_SEC_ROWLEAF2[row] holds the derived-role condition's fused-leaf record */
/* The condition is an all/any/none tree of classified fused leaves (celc.py _tree_strip): the slot holds a descriptor laid
 * over the fused-leaf record's fields - {ops 0-7, ops 8-15, index of the first leaf record in the tape in 8-dword units,
 * ops 16-23, ops 24-31, n leaves, 0, 7} - and the n leaf records follow each other there.  ops = 4 bits each: 1 = the next
 * leaf, 2 + k / 5 + k / 8 + k = OP_TREE_BEGIN / _ACC / _END of kind k (0 all, 1 any, 2 none), 0 = end
 * (cbh_check_flat.h flat_tree; the tape program is unchanged). */
#define CBH_ROW_F_TREE_EMBEDDED 256u
#define CBH_ROW_F_DRTREE_EMBEDDED 512u
#define CBH_TREE_STRIP_MAX 8u
/* cbh_check_walk2.h.  CBH_ROW_F_XEXACT: the record's match is decided exactly by class masks + glob masks (every resource-policy
 * record of a CBH_MF_WALK2 table).  CBH_ROW_F_X: CBH_SEC_ROWX[row] holds something - glob masks (then also the class masks of
 * the LITERAL list entries only; the record's own masks say "every class" for a list with a glob) or evaluation-site slots. */
#define CBH_ROW_F_X 1024u
#define CBH_ROW_F_XEXACT 2048u
#define CBH_ROW_F_OUTPUT 4096u   /* the rule has output expressions: a visit may emit an OutputEntry (check.go:383-411) */
enum CbhRowXField {
  CBH_ROWX_GSLOTS = 0,   // slot of the condition | slot of the derived-role condition << 16 (CBH_GSLOT_NONE = none)
  CBH_ROWX_GLOBS = 1,    // action glob mask | role glob mask << 16 (bit = glob index in the dimension, < CBH_W2_MAX_GLOBS)
  CBH_ROWX_ROLES = 2,    // u64: classes of the literal roles
  CBH_ROWX_ACTIONS = 4,  // u64: classes of the literal actions
  CBH_ROWX_PROBES = 6,   // site slot of the probe of the rule's variables | of its derived-role params' variables << 16 (celc.py
                         // vars_probe_program: did any variable of the params set fail? - evaluated on a visit, check.go:651-677)
  CBH_ROWX_PROBE_PCS = 7, // U32POOL offset of the two probe programs (CBH_NONE each where there is none), or CBH_NONE
  CBH_ROWX_NF = 8
};
#define CBH_SWF_PRINCIPAL 1u    /* the string is a principal with a principal policy (some version / scope) */
#define CBH_SWF_PARENTS 2u      /* the string is a role with ancestors in some scope */
#define CBH_SCOPE_F_ROLEPOL 16u /* CBH_SEC_SCOPE_FLAGS bit 4: some role policy lives at this scope */
#define CBH_BS_ROW_GENERIC 2u
#define CBH_BS_ROW_OPEN 4u
#define CBH_BS_DR_GENERIC 8u
#define CBH_BS_DR_OPEN 16u
#define CBH_GSLOT_NONE 0xFFFFu
#define CBH_W2_MAX_GLOBS 16u
#define CBH_W2_MAX_GSLOTS 256u
#define CBH_W2_SLOTS_PER_WORD 16u   /* four result bits per site: 1 satisfied, 2 CEL error, 8 outside the device subset */
enum CbhRpxField {       // one role-policy rule (same index as CBH_SEC_RPROWS)
  CBH_RPX_RESOURCE = 0,  // pattern ref (kind dim)
  CBH_RPX_CNT = 1,       // CBH_RP_ALLOW_CNT word (count | CBH_RP_F_*)
  CBH_RPX_COND = 2,      // program of the user condition or CBH_NONE
  CBH_RPX_GSLOT = 3,
  CBH_RPX_ACTIONS = 4,   // u64: classes of the literal allow actions
  CBH_RPX_AGLOBS = 6,    // glob mask of the allow list (action dim)
  CBH_RPX_HOW = 7,       // bits 0-1: 1 = dwords 8..15 hold the condition's fused-leaf record, 2 = a tree descriptor (CBH_ROW_F_TREE_EMBEDDED);
                         // bit 2: the rule has output expressions
  CBH_RPX_LEAF = 8,
  CBH_RPX_NF = 16
};
enum CbhRpField { // role-policy rows
  CBH_RP_RESOURCE = 0,  // pattern ref (kind dim)
  CBH_RP_ALLOW_OFF = 1, // into U32POOL: pattern refs (action dim)
  CBH_RP_ALLOW_CNT = 2, // count (CBH_RP_CNT_MASK) | CBH_RP_F_*
  CBH_RP_COND = 3,      // program of the USER condition (synthetic DENY fires when it is false)
  CBH_RP_NF = 4
};
// The evaluation key of a role-policy rule leaves the resource out (ruletable.go:445-455), so rules of one role policy
// for different resources share it, and with it their entry of the per-request conditionCache (check.go:186, 324).  When a
// glob makes two of them match the same resource kind, the first one a request's actions reach decides what the other sees:
//   OUTPUT_ONLY  a rule without a condition but with an output expression: the reference visits it as a binding without
//                effect (index.go:463-484), which caches "satisfied";
//   SHARES_KEY   the rule shares its key with a rule of the other sort (conditional <-> output only).  A conditional rule
//                reached after an output-only one would deny whatever its condition says: the kernels mark those actions
//                CBH_ST_UNSUPPORTED (the order of the request's actions decides, not the policy).  The trace pass gives
//                an output-only rule reached after a conditional one that rule's cached outcome.
// (Two CONDITIONAL rules in that position are history dependent outright: blob.py makes their conditions UNSUPPORTED programs.)
#define CBH_RP_CNT_MASK 0x3FFFFFFFu
#define CBH_RP_F_OUTPUT_ONLY 0x80000000u
#define CBH_RP_F_SHARES_KEY 0x40000000u
enum CbhDrField { // derived roles of one resource policy
  CBH_DR_NAME = 0,        // bit index into edr mask
  CBH_DR_PARENTS_OFF = 1, // into U32POOL: role string ids
  CBH_DR_PARENTS_CNT = 2, // CBH_NONE = "*" (any role)
  CBH_DR_COND = 3,        // program or CBH_NONE
  CBH_DR_NF = 4
};

enum CbhDrxField { // CBH_SEC_DRX: one 16-dword record per CBH_SEC_DR record, same index
  CBH_DRX_ROLES = 0,   // u64 (2 dwords): role classes of the parent roles (every bit for "*"), exact for a flat table
  CBH_DRX_FLAGS = 2,   // bit 0: dwords 8..15 hold the condition's fused-leaf record; bit 1: a tree descriptor (CBH_ROW_F_TREE_EMBEDDED)
  CBH_DRX_COND = 3,    // program or CBH_NONE
  CBH_DRX_NAME = 4,    // bit index into the edr mask
  CBH_DRX_GSLOT = 5,   // evaluation-site slot of the condition (cbh_check_walk2.h), CBH_GSLOT_NONE = none; << 16: slot of the probe of
                       // the definition's variables
  CBH_DRX_PROBE = 6,   // that probe's program, or CBH_NONE
  CBH_DRX_LEAF = 8,
  CBH_DRX_NF = 16
};

// Glob NFA section (bit-parallel, one bit per pattern position):
//   u64 init[NW]; u64 star[NW]; u64 cls[256][NW]; u64 self[256][NW];
//   u32 n_accept; u32 pad; { u32 bitpos; u32 glob_index; } accept[n_accept];
// step(c): A = ((A & cls[c]) << 1) | (A & self[c]); then closure A |= (A & star) << 1 to fixpoint.

// Bytecode: one u32 per instruction = op | (arg << 8); some ops take a second word.
enum CbhOp {
  OP_RET = 0,
  OP_CONST = 1,      // push const[arg]
  OP_COL = 2,        // push column arg (ABSENT -> error)
  OP_HASCOL = 3,     // push presence of column arg
  OP_REQSTR = 4,     // push request string field arg (cbh_req_field)
  OP_ROLES = 5,      // push P.roles
  OP_SELECT = 6,     // TOS = TOS.<string id arg>
  OP_HASSEL = 7,     // TOS = has(TOS.<string id arg>)
  OP_INDEX = 8,      // pop i; TOS = TOS[i]
  OP_EQ = 9, OP_NE = 10, OP_LT = 11, OP_LE = 12, OP_GT = 13, OP_GE = 14,
  OP_IN = 15,
  OP_ADD = 16, OP_SUB = 17, OP_MUL = 18, OP_DIV = 19, OP_MOD = 20,
  OP_NEG = 21, OP_NOT = 22,
  OP_JF = 23,        // if TOS is bool false: pc = arg (TOS kept)
  OP_JT = 24,        // if TOS is bool true : pc = arg (TOS kept)
  OP_AND = 25, OP_OR = 26, // pop b, a; push a && b / a || b with CEL error absorption
  OP_JTERN = 27,     // pop c; non-bool/error: push error, pc = next word; false: pc = arg; true: skip next word
  OP_JMP = 28,
  OP_POP = 29,
  OP_LEAF = 30,      // TOS -> plain bool (error/non-bool -> false; strict mode + error -> abort)
  OP_SIZE = 31, OP_STARTSWITH = 32, OP_ENDSWITH = 33, OP_CONTAINS = 34,
  OP_TIMESTAMP = 35, OP_DURATION = 36, OP_TIMESINCE = 37, OP_NOW = 38,
  OP_EDRHAS = 39,    // push (derived role bit arg) in runtime.effectiveDerivedRoles
  OP_LOCAL = 40,     // push local arg
  OP_ITER_BEGIN = 41, // pop container -> iteration slot arg ; next word = kind | (end_pc << 8)
  OP_ITER_NEXT = 42,  // arg = slot; next word = end_pc : exhausted -> pc = end_pc, else bind locals
  OP_ITER_ACC = 43,   // arg = slot; next word = loop_pc : pop predicate, fold, maybe finish
  OP_ITER_END = 44,   // arg = slot: push folded result
  OP_TOINT = 45 /* arg 1: uint(x) */, OP_TODOUBLE = 46, OP_TOSTRING_UNSUPPORTED = 47,
  OP_INIPRANGE = 48, // pop cidr, ip (strings)
  OP_UNSUPPORTED = 49, // marks the tuple CBH_ST_UNSUPPORTED and yields an error
  OP_TS_GETTER = 50,  // arg = getter kind; pops tz string if arg bit 7 set
  OP_HASINTERSECTION = 51, OP_ISSUBSET = 52,
  OP_LEAF_BIN = 53,   // arg = binop | kindA << 8 | kindB << 12; next two words = operand args; pushes a plain bool
  OP_TERN = 54,       // pop else, then, guard -> guard ? then : else (error if the guard is not a bool)
  OP_TREE_BEGIN = 55, // arg = kind (0 all, 1 any, 2 none): open a condition tree level
  OP_TREE_ACC = 56,   // arg = kind: pop a child's plain-bool result into the level's accumulator
  OP_TREE_END = 57,   // arg = kind: close the level, push its result
  OP_INDEXOF = 60,    // arg 0 first / 1 last: pop sub, s (strings) -> code-point index of the occurrence, -1 without one
  OP_STREQ_CASE = 61, // arg = mode a | mode b << 2 | ne << 4 (mode 1 lowerAscii, 2 upperAscii): pop b, a -> a' == b' without building a' / b'
  OP_MATCHES = 59,    // next word = offset of the pattern's tables in CBH_SEC_REGEX: TOS (string) -> RE2 MatchString
  OP_HIER = 58,       // arg = predicate (0 ancestorOf, 1 descendentOf, 2 immediateParentOf, 3 immediateChildOf, 4 siblingOf,
                      // 5 overlaps): pop b, a (dot-delimited strings) -> hierarchy(a).<predicate>(hierarchy(b))
  OP_VARSCOPE = 62,   // TOS is the inlined definition of the variable named by trace string <arg & 0x7FFFFF>.  Mode (arg >> 23) 0,
                      // trace programs only: an error there left the variable unset (check.go:651-677), the reference reads
                      // "undefined field '<name>'".  Mode 1, variables of a derived-role definition (check.go:612-633): the
                      // error is recorded and the variable is null (unset in strict mode)
  OP_OUT = 63,        // trace programs: TOS is the value of an output expression (check.go:776-807): logged, not returned
  OP_LISTOP = 64,     // arg 0 intersect / 1 except / 2 concatenation: pop b, a (lists) -> a new list in the lane's arena
  OP_STRCAT = 66,     // pop b, a (strings or ropes) -> the rope a ++ b: its parts side by side in the lane's arena, no byte is copied
  OP_STRCASE = 67,    // arg 1 lowerAscii / 2 upperAscii: TOS string -> the rope that reads it through the case mapping
  OP_LISTFN = 65,     // arg 0 reverse: TOS list -> reversed copy in the arena; 1 slice: pop end, start; TOS list -> the view
                      // [start, end) of it; 2 lists.range: TOS int n -> [0 .. n) in the arena
  OP_IPFN = 68,       // cel-go ext.Network on a string the request supplies.  arg 0 isIP(s), 8 / 9 isIP(s, 4 / 6), 7 ip.isCanonical(s); 1 ip(s).family(),
                      // 2 isUnspecified, 3 isLoopback, 4 isLinkLocalUnicast, 5 isLinkLocalMulticast, 6 isGlobalUnicast (an error where
                      // ip(s) fails); 10: pop ip, cidr (strings) -> cidr(c).containsIP(ip)
  OP_STRVIEW = 69,    // arg 0 substring(a): pop a; 1 substring(a, b): pop b, a; 2 charAt(i): pop i; 3 trim().  TOS string -> the rope that
                      // is that window of it (code-point indices, cel-go ext/strings.go)
  OP_STRREPLACE = 70, // pop new, old; TOS string s -> the rope s.replace(old, new): the pieces of s between the occurrences, `new` between them
  OP_EDREQ = 72,      // arg = constant index of a 64-bit mask | never << 31: push runtime.effectiveDerivedRoles == <the constant list whose names
                      // the mask holds> (the runtime list is sorted and duplicate free: a constant list that is not can never equal it)
  OP_HIERCOMMON = 71, // pop c, b; TOS a (dot-delimited strings) -> hierarchy(a).commonAncestors(hierarchy(b)) == hierarchy(c)
  OP_EDRVAL = 73,     // trace programs only: push runtime.effectiveDerivedRoles AS A VALUE (CBH_T_EDRSET, the mask of the scope being walked) -
                      // emitted only where it is the WHOLE expression of an output's part or of a variable: nothing else ever sees the tag
  OP_NOPS
};
enum CbhIterKind { IT_ALL = 0, IT_EXISTS = 1, IT_EXISTS_ONE = 2, IT_FILTER = 3, IT_MAP = 4,   // filter / map build a list in the lane's arena
                   IT_MAP_FILTER = 5 };   // map(x, pred, expr) / transformList(i, v, pred, expr): the body leaves pred and expr
#define CBH_ARENA_ENTRIES 48u   /* values per lane a program may build lists from; more marks the tuple CBH_ST_UNSUPPORTED */
