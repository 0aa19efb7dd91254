// frontend.cpp -- hand-written .csp front end: tokenizer, recursive-descent parser, model
// build and normalisation, flattened to the stcsp_problem POD of include/stcsp_engine.h.
//
// Behavioural mirror (not a transcription) of
//   src/stcsp.l                 token set, longest-match rules, comment forms
//   src/stcsp.y:56-174          grammar, precedence, AST shapes
//   src/solver.cpp:110-159      solverAddVar / solverAuxVarNew / solverAddArr / solverParse
//   src/solveralgorithm.cpp:16-332  solverAddConstr, constraintNormalise (aux-variable
//                               introduction for next / fby / @, expression bounds)
//   src/constraint.cpp:58-89    constraintNodeParse (name resolution)
// There is no lex/yacc in the image, and the build must not depend on generators anyway.
//
// This is host plumbing in front of the engine seam; nothing here runs per search node.
#include <cctype>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cvalue.hpp"
#include "stcsp_host.h"
#include "tree.hpp"

namespace stcsp {
static inline int neg_wrap(int x) { return (int)(0u - (unsigned)x); }  // two's-complement negation without signed overflow


static thread_local std::string g_last_error;

static void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

struct FrontendError {
    int code;
};

// ------------------------------------------------------------------ tokens
enum Tok {
    TK_EOF = 0,
    TK_IDENT = 256,
    TK_CONST,
    TK_VAR,
    TK_OBJ,
    TK_ARR,
    TK_LE_CON,
    TK_GE_CON,
    TK_EQ_CON,
    TK_NE_CON,
    TK_IMPLY_CON,
    TK_UNTIL_CON,
    TK_LT_OP,
    TK_GT_OP,
    TK_LE_OP,
    TK_GE_OP,
    TK_EQ_OP,
    TK_NE_OP,
    TK_AND,
    TK_OR,
    TK_NOT,
    TK_AT,
    TK_ABS,
    TK_FIRST,
    TK_NEXT,
    TK_FBY,
    TK_IF,
    TK_THEN,
    TK_ELSE
};

struct Token {
    int kind;  // Tok or a literal character
    std::string text;
    int num;
    int line;
};

// stcsp.l:23-62 keyword table. Keywords only win at equal length (flex longest match), so an
// identifier is scanned first and then looked up.
static int keyword(const std::string &s) {
    static const struct {
        const char *k;
        int t;
    } table[] = {{"var", TK_VAR},     {"obj", TK_OBJ},     {"arr", TK_ARR},   {"until", TK_UNTIL_CON},
                 {"lt", TK_LT_OP},    {"gt", TK_GT_OP},    {"le", TK_LE_OP},  {"ge", TK_GE_OP},
                 {"eq", TK_EQ_OP},    {"ne", TK_NE_OP},    {"and", TK_AND},   {"or", TK_OR},
                 {"not", TK_NOT},     {"abs", TK_ABS},     {"first", TK_FIRST}, {"next", TK_NEXT},
                 {"fby", TK_FBY},     {"if", TK_IF},       {"then", TK_THEN}, {"else", TK_ELSE}};
    for (auto &e : table)
        if (s == e.k) return e.t;
    return 0;
}

static std::vector<Token> tokenize(const std::string &src) {
    std::vector<Token> out;
    size_t i = 0, n = src.size();
    int line = 1;
    auto push = [&](int kind, const std::string &text = std::string(), int num = 0) {
        out.push_back(Token{kind, text, num, line});
    };
    while (i < n) {
        unsigned char c = (unsigned char)src[i];
        // "//"[^\n]*\n  : the newline is part of the match, line_num is NOT bumped (stcsp.l:20)
        if (c == '/' && i + 1 < n && src[i + 1] == '/') {
            size_t j = i + 2;
            while (j < n && src[j] != '\n') j++;
            if (j < n) {
                i = j + 1;
                continue;
            }
            // no trailing newline: the rule does not match; '/' '/' lex as operators
        }
        // "/*"[^"*/"]*"*/" : the body may not contain '"', '*' or '/' (stcsp.l:21)
        if (c == '/' && i + 1 < n && src[i + 1] == '*') {
            size_t j = i + 2;
            while (j < n && src[j] != '"' && src[j] != '*' && src[j] != '/') {
                if (src[j] == '\n') {}  // newlines inside do not bump line_num either
                j++;
            }
            if (j + 1 < n && src[j] == '*' && src[j + 1] == '/') {
                i = j + 2;
                continue;
            }
        }
        if (c == '\'') {  // "'"[^\n]*  (stcsp.l:79)
            while (i < n && src[i] != '\n') i++;
            continue;
        }
        if (c == '\n') {
            line++;
            i++;
            continue;
        }
        if (c == ' ' || c == '\t' || c == '\v' || c == '\f') {
            i++;
            continue;
        }
        if (isalpha(c)) {  // {L}({L}|{D})*
            size_t j = i + 1;
            while (j < n && isalnum((unsigned char)src[j])) j++;
            std::string s = src.substr(i, j - i);
            int k = keyword(s);
            if (k)
                push(k);
            else
                push(TK_IDENT, s);
            i = j;
            continue;
        }
        // [-]?{D}+ beats "-" and "->" only when it is the longer match
        if (isdigit(c) || (c == '-' && i + 1 < n && isdigit((unsigned char)src[i + 1]))) {
            size_t j = i + 1;
            while (j < n && isdigit((unsigned char)src[j])) j++;
            push(TK_CONST, std::string(), atoi(src.substr(i, j - i).c_str()));
            i = j;
            continue;
        }
        if (i + 1 < n) {
            char d = src[i + 1];
            int two = 0;
            if (c == '<' && d == '=') two = TK_LE_CON;
            if (c == '>' && d == '=') two = TK_GE_CON;
            if (c == '=' && d == '=') two = TK_EQ_CON;
            if (c == '!' && d == '=') two = TK_NE_CON;
            if (c == '-' && d == '>') two = TK_IMPLY_CON;
            if (two) {
                push(two);
                i += 2;
                continue;
            }
        }
        if (c == '@') {
            push(TK_AT);
            i++;
            continue;
        }
        push((int)c);  // single-character operators / punctuation / stray characters
        i++;
    }
    push(TK_EOF);
    return out;
}

// ------------------------------------------------------------------ AST (role of node.h:5-12)
struct Ast {
    int kind;  // AK_*
    std::string str;
    int num1 = 0, num2 = 0;
    Ast *left = nullptr, *right = nullptr;
    std::vector<int> list;  // array literal, in source order
};
enum AstKind { AK_VARDECL = 1, AK_ARRDECL, AK_OBJ, AK_CONSTRAINT_OR_EXPR };

struct Parser {
    std::vector<Token> toks;
    size_t pos = 0;
    std::vector<std::unique_ptr<Ast>> pool;
    struct Stmt {
        int kind;
        std::string name;
        int lo = 0, hi = 0;
        std::vector<int> elems;
        Ast *expr = nullptr;
    };

    Ast *mk(int token, Ast *l = nullptr, Ast *r = nullptr) {
        pool.emplace_back(new Ast());
        Ast *a = pool.back().get();
        a->kind = token;
        a->left = l;
        a->right = r;
        return a;
    }
    int peek() const { return toks[pos].kind; }
    const Token &take() { return toks[pos++]; }
    [[noreturn]] void syntax_error() {
        // yyerror (stcsp.y:221-224) prints "Line %d: %s\n" to stdout and exits 1
        set_error("Line %d: syntax error", toks[pos < toks.size() ? pos : toks.size() - 1].line);
        throw FrontendError{STCSP_E_INVALID};
    }
    const Token &expect(int k) {
        if (peek() != k) syntax_error();
        return take();
    }

    // precedence ladder of stcsp.y:101-174 (low -> high):
    //   not < or < and < eq ne < lt gt le ge < + - < * / % < @ CONST < fby (right assoc)
    //   < unary first/next/abs/if-then-else < primary
    Ast *expression() { return logical_not(); }
    Ast *logical_not() {
        if (peek() == TK_NOT) {
            take();
            return mk(STCSP_T_NOT, nullptr, logical_not());
        }
        return logical_or();
    }
    Ast *logical_or() {
        Ast *l = logical_and();
        while (peek() == TK_OR) {
            take();
            l = mk(STCSP_T_OR, l, logical_and());
        }
        return l;
    }
    Ast *logical_and() {
        Ast *l = equality();
        while (peek() == TK_AND) {
            take();
            l = mk(STCSP_T_AND, l, equality());
        }
        return l;
    }
    Ast *equality() {
        Ast *l = relational();
        while (peek() == TK_EQ_OP || peek() == TK_NE_OP) {
            int t = take().kind == TK_EQ_OP ? STCSP_T_EQ_OP : STCSP_T_NE_OP;
            l = mk(t, l, relational());
        }
        return l;
    }
    Ast *relational() {
        Ast *l = additive();
        for (;;) {
            int t = 0;
            switch (peek()) {
                case TK_LT_OP: t = STCSP_T_LT_OP; break;
                case TK_GT_OP: t = STCSP_T_GT_OP; break;
                case TK_LE_OP: t = STCSP_T_LE_OP; break;
                case TK_GE_OP: t = STCSP_T_GE_OP; break;
                default: break;
            }
            if (!t) return l;
            take();
            l = mk(t, l, additive());
        }
    }
    Ast *additive() {
        Ast *l = multiplicative();
        while (peek() == '+' || peek() == '-') {
            int t = take().kind == '+' ? STCSP_T_ADD : STCSP_T_SUB;
            l = mk(t, l, multiplicative());
        }
        return l;
    }
    Ast *multiplicative() {
        Ast *l = at_expr();
        while (peek() == '*' || peek() == '/' || peek() == '%') {
            int k = take().kind;
            int t = k == '*' ? STCSP_T_MUL : (k == '/' ? STCSP_T_DIV : STCSP_T_MOD);
            l = mk(t, l, at_expr());
        }
        return l;
    }
    Ast *at_expr() {  // fby_expression [ AT CONSTANT ]   (not repeatable, stcsp.y:151-154)
        Ast *l = fby_expr();
        if (peek() == TK_AT) {
            take();
            const Token &c = expect(TK_CONST);
            Ast *a = mk(STCSP_T_AT, l, nullptr);
            a->num1 = c.num;
            return a;
        }
        return l;
    }
    Ast *fby_expr() {  // right associative (stcsp.y:156-159)
        Ast *l = unary();
        if (peek() == TK_FBY) {
            take();
            return mk(STCSP_T_FBY, l, fby_expr());
        }
        return l;
    }
    Ast *unary() {
        switch (peek()) {
            case TK_FIRST: take(); return mk(STCSP_T_FIRST, nullptr, unary());
            case TK_NEXT: take(); return mk(STCSP_T_NEXT, nullptr, unary());
            case TK_ABS: take(); return mk(STCSP_T_ABS, nullptr, unary());
            case TK_IF: {
                take();
                Ast *c = expression();
                expect(TK_THEN);
                Ast *a = expression();
                expect(TK_ELSE);
                Ast *b = unary();  // the else arm is a unary_expression (stcsp.y:165)
                return mk(STCSP_T_IF, c, mk(STCSP_T_THEN, a, b));
            }
            default: return primary();
        }
    }
    Ast *primary() {
        if (peek() == TK_IDENT) {
            const Token &t = take();
            if (peek() == '[') {
                take();
                Ast *e = expression();
                expect(']');
                Ast *a = mk(STCSP_T_ARR, nullptr, e);
                a->str = t.text;
                return a;
            }
            Ast *a = mk(STCSP_T_VAR);
            a->str = t.text;
            return a;
        }
        if (peek() == TK_CONST) {
            Ast *a = mk(STCSP_T_CONST);
            a->num1 = take().num;
            return a;
        }
        if (peek() == '(') {
            take();
            Ast *e = expression();
            expect(')');
            return e;
        }
        syntax_error();
    }

    Stmt statement() {
        Stmt s;
        if (peek() == TK_VAR) {  // VAR IDENTIFIER ':' '[' CONSTANT ',' CONSTANT ']' ';'
            take();
            s.kind = AK_VARDECL;
            s.name = expect(TK_IDENT).text;
            expect(':');
            expect('[');
            s.lo = expect(TK_CONST).num;
            expect(',');
            s.hi = expect(TK_CONST).num;
            expect(']');
            expect(';');
            return s;
        }
        if (peek() == TK_ARR) {  // ARR IDENTIFIER ':' '{' c {',' c} '}' ';'
            take();
            s.kind = AK_ARRDECL;
            s.name = expect(TK_IDENT).text;
            expect(':');
            expect('{');
            s.elems.push_back(expect(TK_CONST).num);
            while (peek() == ',') {
                take();
                s.elems.push_back(expect(TK_CONST).num);
            }
            expect('}');
            expect(';');
            return s;
        }
        if (peek() == TK_OBJ) {  // parses, but is rejected downstream (solver.cpp:154-156)
            take();
            s.kind = AK_OBJ;
            s.name = expect(TK_IDENT).text;
            expect(';');
            return s;
        }
        s.kind = AK_CONSTRAINT_OR_EXPR;
        Ast *l = expression();
        int t = 0;
        switch (peek()) {
            case '<': t = STCSP_T_LT_CON; break;
            case '>': t = STCSP_T_GT_CON; break;
            case TK_LE_CON: t = STCSP_T_LE_CON; break;
            case TK_GE_CON: t = STCSP_T_GE_CON; break;
            case TK_EQ_CON: t = STCSP_T_EQ_CON; break;
            case TK_NE_CON: t = STCSP_T_NE_CON; break;
            case TK_UNTIL_CON: t = STCSP_T_UNTIL_CON; break;
            case TK_IMPLY_CON: t = STCSP_T_IMPLY_CON; break;
            default: syntax_error();
        }
        take();
        Ast *r = expression();
        expect(';');
        s.expr = mk(t, l, r);
        return s;
    }
    std::vector<Stmt> program() {
        std::vector<Stmt> out;
        while (peek() != TK_EOF) out.push_back(statement());
        return out;
    }
};

// ------------------------------------------------------------------ model build + normalise
struct VarDecl {
    std::string name;
    int lb, ub;
};

struct Model {
    int prefix_k = 2;
    std::vector<VarDecl> vars;
    std::vector<std::string> array_names;
    ArrayTable arrays;
    std::vector<Tree *> constraints;  // constrQueue order
    int num_aux = 0;
    TreeArena arena;

    // flattened view
    std::vector<int32_t> f_lb, f_ub, f_aoff, f_adata, f_roots;
    std::vector<const char *> f_names;
    std::vector<stcsp_node> f_nodes;
    stcsp_problem problem{};

    int find_var(const std::string &n) const {  // solverGetVar: first match wins
        for (size_t i = 0; i < vars.size(); i++)
            if (vars[i].name == n) return (int)i;
        return -1;
    }
    int find_array(const std::string &n) const {
        for (size_t i = 0; i < array_names.size(); i++)
            if (array_names[i] == n) return (int)i;
        return -1;
    }
    int add_var(const std::string &name, int lb, int ub) {  // variableNew (variable.cpp:10-43)
        if (lb > ub) {
            set_error("Invalid domain [%d, %d] in variable %s", lb, ub, name.c_str());
            throw FrontendError{STCSP_E_INVALID};
        }
        vars.push_back(VarDecl{name, lb, ub});
        return (int)vars.size() - 1;
    }
    int aux_var(int lb, int ub) {  // solverAuxVarNew (solver.cpp:119-124)
        char name[32];
        snprintf(name, sizeof name, "_V%d", num_aux++);
        return add_var(name, lb, ub);
    }
    // solverAddConstrNode (solver.cpp:133-136): pushed as is -- no tautology test, no normalise
    void push_constraint(Tree *root) { constraints.push_back(root); }

    // constraintNodeParse (constraint.cpp:58-89)
    Tree *resolve(const Ast *a) {
        if (!a) return nullptr;
        switch (a->kind) {
            case STCSP_T_CONST: return arena.constant(a->num1);
            case STCSP_T_VAR: {
                int v = find_var(a->str);
                if (v < 0) {
                    set_error("Variable '%s' has not been defined.", a->str.c_str());
                    throw FrontendError{STCSP_E_INVALID};
                }
                return arena.variable(v);
            }
            case STCSP_T_ARR: {
                int ar = find_array(a->str);
                if (ar < 0) {
                    set_error("Variable '%s' has not been defined.", a->str.c_str());
                    throw FrontendError{STCSP_E_INVALID};
                }
                return arena.make(STCSP_T_ARR, 0, -1, ar, nullptr, resolve(a->right));
            }
            case STCSP_T_AT: return arena.make(STCSP_T_AT, 0, -1, -1, resolve(a->left), arena.constant(a->num1));
            case STCSP_T_ABS:
            case STCSP_T_FIRST:
            case STCSP_T_NEXT:
            case STCSP_T_NOT: return arena.make(a->kind, 0, -1, -1, nullptr, resolve(a->right));
            default: {
                Tree *l = resolve(a->left);  // left before right: errors surface in source order
                Tree *r = resolve(a->right);
                return arena.make(a->kind, 0, -1, -1, l, r);
            }
        }
    }

    void add_eq(Tree *l, Tree *r) { push_constraint(arena.make(STCSP_T_EQ_CON, 0, -1, -1, l, r)); }
    // solverAddConstrVarEqNext / FirstEqFirst / VarEqNode / VarEqAt (solveralgorithm.cpp:27-57)
    void add_var_eq_next(int x, int y) {
        add_eq(arena.variable(x), arena.make(STCSP_T_NEXT, 0, -1, -1, nullptr, arena.variable(y)));
    }
    void add_first_eq_first(int x, int y) {
        add_eq(arena.make(STCSP_T_FIRST, 0, -1, -1, nullptr, arena.variable(x)),
               arena.make(STCSP_T_FIRST, 0, -1, -1, nullptr, arena.variable(y)));
    }
    void add_var_eq_node(int x, Tree *n) { add_eq(arena.variable(x), n); }
    void add_var_eq_at(int x, int y, int k) {
        add_eq(arena.variable(x), arena.make(STCSP_T_AT, 0, -1, -1, arena.variable(y), arena.constant(k)));
    }

    // constraintNormalise (solveralgorithm.cpp:60-332). Rewrites next / fby / @ into auxiliary
    // variables + primitive constraints and computes expression bounds. Aux constraints are
    // pushed BEFORE the statement's own constraint; aux variables are appended in creation
    // order. lb/ub are left untouched where the reference leaves them unassigned, except that
    // they start from a defined value here (see `not` below).
    Tree *normalise(Tree *node, int &lb, int &ub) {
        if (!node) return nullptr;
        int llo = 0, lhi = 1, rlo = 0, rhi = 1;
        switch (node->token) {
            case STCSP_T_FIRST: {
                Tree *r = node->right;
                if (r->token == STCSP_T_CONST) {  // first c -> c            (:79-83)
                    lb = ub = r->num;
                    return r;
                }
                if (r->token == STCSP_T_VAR) {  // first x stays              (:84-87)
                    lb = vars[r->var].lb;
                    ub = vars[r->var].ub;
                    return node;
                }
                if (r->token == STCSP_T_FBY) {  // first (a fby b) -> first a (:104-109)
                    node->right = r->left;
                    return normalise(node, lb, ub);
                }
                Tree *nr = normalise(r, llo, lhi);  // first e -> first e', nested first stripped (:110-121)
                if (nr->token == STCSP_T_FIRST) nr = nr->right;
                node->right = nr;
                lb = llo;
                ub = lhi;
                return node;
            }
            case STCSP_T_NEXT: {
                Tree *r = node->right;
                if (r->token == STCSP_T_CONST) {  // next c -> c              (:123-127)
                    lb = ub = r->num;
                    return r;
                }
                if (r->token == STCSP_T_VAR) {  // next x -> aux v, v == next x (:128-135)
                    lb = vars[r->var].lb;
                    ub = vars[r->var].ub;
                    int x = aux_var(lb, ub);
                    add_var_eq_next(x, r->var);
                    return arena.variable(x);
                }
                if (r->token == STCSP_T_FBY)  // next (a fby b) -> b          (:136-141)
                    return normalise(r->right, lb, ub);
                Tree *nr = normalise(r, llo, lhi);  //                     (:142-160)
                Tree *res;
                if (nr->token == STCSP_T_FIRST) {
                    res = nr;  // the `next` is dropped
                } else if (nr->token == STCSP_T_CONST || nr->token == STCSP_T_VAR) {
                    node->right = nr;
                    res = normalise(node, llo, lhi);
                } else {
                    int x = aux_var(llo, lhi);
                    add_var_eq_node(x, nr);
                    int y = aux_var(llo, lhi);
                    add_var_eq_next(y, x);
                    res = arena.variable(y);
                }
                lb = llo;
                ub = lhi;
                return res;
            }
            case STCSP_T_FBY: {  // (:161-188)
                int y, z;
                if (node->left->token == STCSP_T_VAR) {
                    y = node->left->var;
                    llo = vars[y].lb;
                    lhi = vars[y].ub;
                } else {
                    Tree *l = normalise(node->left, llo, lhi);
                    y = aux_var(llo, lhi);
                    add_var_eq_node(y, l);
                }
                if (node->right->token == STCSP_T_VAR) {
                    z = node->right->var;
                    rlo = vars[z].lb;
                    rhi = vars[z].ub;
                } else {
                    Tree *r = normalise(node->right, rlo, rhi);
                    z = aux_var(rlo, rhi);
                    add_var_eq_node(z, r);
                }
                lb = llo < rlo ? llo : rlo;
                ub = lhi > rhi ? lhi : rhi;
                int x = aux_var(lb, ub);
                add_first_eq_first(x, y);
                add_var_eq_next(z, x);
                return arena.variable(x);
            }
            case STCSP_T_AT: {  // (:189-227)
                Tree *l = node->left;
                if (l->token == STCSP_T_CONST) {
                    lb = ub = l->num;
                    return l;
                }
                if (l->token == STCSP_T_NEXT) {  // (next^n e) @ k -> e @ (k+n); bounds untouched
                    int n = 0;
                    while (node->left && node->left->token == STCSP_T_NEXT) {
                        n++;
                        node->left = node->left->right;
                    }
                    node->right->num += n;
                    return node;
                }
                int y;
                if (l->token == STCSP_T_VAR) {
                    y = l->var;
                    llo = vars[y].lb;
                    lhi = vars[y].ub;
                } else {
                    Tree *nl = normalise(l, llo, lhi);
                    y = aux_var(llo, lhi);
                    add_var_eq_node(y, nl);
                }
                lb = llo;
                ub = lhi;
                int x = aux_var(lb, ub);
                add_var_eq_at(x, y, node->right->num);
                return arena.variable(x);
            }
            case STCSP_T_VAR:
                lb = vars[node->var].lb;
                ub = vars[node->var].ub;
                return node;
            case STCSP_T_CONST: lb = ub = node->num; return node;
            case STCSP_T_ARR: {  // (:237-247) bounds = min/max element
                node->right = normalise(node->right, llo, lhi);
                const std::vector<int> &e = arrays.elements[node->arr];
                lb = ub = e[0];
                for (int v : e) {
                    if (v < lb) lb = v;
                    if (v > ub) ub = v;
                }
                return node;
            }
            case STCSP_T_UNTIL_CON: {  // (:248-261) both sides forced to identifiers
                bool left_is_var = node->left->token == STCSP_T_VAR;    // the reference tests the
                bool right_is_var = node->right->token == STCSP_T_VAR;  // ORIGINAL child tokens
                Tree *l = normalise(node->left, llo, lhi);
                Tree *r = normalise(node->right, rlo, rhi);
                if (!left_is_var) {
                    int x = aux_var(0, 1);
                    node->left = arena.variable(x);
                    add_var_eq_node(x, l);
                }
                if (!right_is_var) {
                    int y = aux_var(0, 1);
                    node->right = arena.variable(y);
                    add_var_eq_node(y, r);
                }
                return node;
            }
            default: break;
        }
        // generic binary / unary branch (:268-327). `not` lands here too: the reference's
        // dedicated `not` arm is dead code (guard typo at :262), so its bounds stay whatever
        // the caller had; here they default to [0,1].
        Tree *l = normalise(node->left, llo, lhi);
        Tree *r = normalise(node->right, rlo, rhi);
        switch (node->token) {
            case STCSP_T_ABS:
                // (wrapping negation: the bounds of `/` and `%` sub-terms are [INT_MIN, INT_MAX], and -INT_MIN is INT_MIN in
                // the reference's build as well, solveralgorithm.cpp:316-322)
                if (rlo < 0 && rhi < 0) {
                    lb = neg_wrap(rhi);
                    ub = neg_wrap(rlo);
                } else if (rlo < 0 && rhi > 0) {
                    lb = 0;
                    ub = (neg_wrap(rlo) > rhi) ? neg_wrap(rlo) : rhi;
                } else {
                    lb = rlo;
                    ub = rhi;
                }
                break;
            case STCSP_T_IF:
                lb = rlo;
                ub = rhi;
                break;
            case STCSP_T_THEN:
                lb = llo < rlo ? llo : rlo;
                ub = lhi > rhi ? lhi : rhi;
                break;
            case STCSP_T_LT_CON: case STCSP_T_GT_CON: case STCSP_T_LE_CON: case STCSP_T_GE_CON:
            case STCSP_T_EQ_CON: case STCSP_T_NE_CON: case STCSP_T_IMPLY_CON:
            case STCSP_T_LT_OP: case STCSP_T_GT_OP: case STCSP_T_LE_OP: case STCSP_T_GE_OP:
            case STCSP_T_EQ_OP: case STCSP_T_NE_OP: case STCSP_T_AND: case STCSP_T_OR:
            case STCSP_T_NOT:
                lb = 0;
                ub = 1;
                break;
            case STCSP_T_ADD:
                lb = (int)((unsigned)llo + (unsigned)rlo);
                ub = (int)((unsigned)lhi + (unsigned)rhi);
                break;
            case STCSP_T_SUB:
                lb = (int)((unsigned)llo - (unsigned)rhi);
                ub = (int)((unsigned)lhi - (unsigned)rlo);
                break;
            case STCSP_T_MUL: {
                auto mul = [](int a, int b) { return (int)((unsigned)a * (unsigned)b); };
                if (llo >= 0 && rlo >= 0) {
                    lb = mul(llo, rlo);
                    ub = mul(lhi, rhi);
                } else if (llo >= 0 && rhi >= 0 && rlo < 0) {
                    lb = mul(lhi, rlo);
                    ub = mul(lhi, rhi);
                } else if (lhi >= 0 && llo < 0 && rlo >= 0) {
                    lb = mul(llo, rlo);
                    ub = mul(lhi, rhi);
                } else {
                    lb = mul(lhi, rhi);
                    ub = mul(llo, rlo);
                }
                break;
            }
            case STCSP_T_DIV:
            case STCSP_T_MOD:
                lb = INT_MIN;
                ub = INT_MAX;
                break;
            default: break;
        }
        node->left = l;
        node->right = r;
        return node;
    }

    // solverAddConstr (solveralgorithm.cpp:16-24)
    void add_constraint(const Ast *a) {
        if (!(a->kind >= STCSP_T_CONST && a->kind <= STCSP_T_UNTIL_CON)) {
            set_error("Unknown token: %d", a->kind);
            throw FrontendError{STCSP_E_INVALID};
        }
        int lb = 0, ub = 0;
        Tree *root = normalise(resolve(a), lb, ub);
        if (!is_tautology(root, arrays)) push_constraint(root);
    }

    void finish() {
        size_t n = vars.size();
        f_lb.resize(n);
        f_ub.resize(n);
        f_names.resize(n);
        for (size_t i = 0; i < n; i++) {
            f_lb[i] = vars[i].lb;
            f_ub[i] = vars[i].ub;
            f_names[i] = vars[i].name.c_str();
        }
        f_aoff.assign(1, 0);
        for (auto &e : arrays.elements) {
            f_adata.insert(f_adata.end(), e.begin(), e.end());
            f_aoff.push_back((int32_t)f_adata.size());
        }
        for (Tree *t : constraints) f_roots.push_back(flatten(t, f_nodes));
        problem.n_vars = (int32_t)n;
        problem.prefix_k = prefix_k;
        problem.var_lb = f_lb.data();
        problem.var_ub = f_ub.data();
        problem.var_names = f_names.data();
        problem.n_arrays = (int32_t)arrays.elements.size();
        problem.array_off = f_aoff.data();
        problem.array_data = f_adata.data();
        problem.n_nodes = (int32_t)f_nodes.size();
        problem.nodes = f_nodes.data();
        problem.n_constraints = (int32_t)f_roots.size();
        problem.constraint_root = f_roots.data();
    }
};

static int build_model(const std::string &text, int prefix_k, Model **out) {
    std::unique_ptr<Model> m(new Model());
    m->prefix_k = prefix_k > 0 ? prefix_k : 2;
    try {
        Parser p;
        p.toks = tokenize(text);
        std::vector<Parser::Stmt> prog = p.program();  // whole file parses before anything is built
        for (auto &s : prog) {                           // solverParse (solver.cpp:138-159)
            switch (s.kind) {
                case AK_VARDECL: m->add_var(s.name, s.lo, s.hi); break;
                case AK_ARRDECL:
                    m->array_names.push_back(s.name);
                    m->arrays.elements.push_back(s.elems);
                    break;
                case AK_OBJ:  // treated as a constraint -> "Unknown token" -> exit(1)
                    set_error("Unknown token: obj");
                    throw FrontendError{STCSP_E_UNSUPPORTED};
                default: m->add_constraint(s.expr); break;
            }
        }
    } catch (const FrontendError &e) {
        return e.code;
    }
    m->finish();
    *out = m.release();
    return STCSP_OK;
}

// ------------------------------------------------------------------ diagnostics printer
static const char *token_text(int t) {
    switch (t) {
        case STCSP_T_AT: return "@";
        case STCSP_T_ABS: return "abs";
        case STCSP_T_NOT: return "not";
        case STCSP_T_FIRST: return "first";
        case STCSP_T_NEXT: return "next";
        case STCSP_T_FBY: return "fby";
        case STCSP_T_AND: return "and";
        case STCSP_T_OR: return "or";
        case STCSP_T_ADD: return "+";
        case STCSP_T_SUB: return "-";
        case STCSP_T_MUL: return "*";
        case STCSP_T_DIV: return "/";
        case STCSP_T_MOD: return "%";
        case STCSP_T_LT_OP: return "lt";
        case STCSP_T_GT_OP: return "gt";
        case STCSP_T_LE_OP: return "le";
        case STCSP_T_GE_OP: return "ge";
        case STCSP_T_EQ_OP: return "eq";
        case STCSP_T_NE_OP: return "ne";
        case STCSP_T_LT_CON: return "<";
        case STCSP_T_GT_CON: return ">";
        case STCSP_T_LE_CON: return "<=";
        case STCSP_T_GE_CON: return ">=";
        case STCSP_T_EQ_CON: return "==";
        case STCSP_T_NE_CON: return "!=";
        case STCSP_T_IMPLY_CON: return "->";
        case STCSP_T_UNTIL_CON: return "until";
        default: return "?";
    }
}

static void print_tree(const Model &m, const Tree *t, std::string &out) {
    if (!t) return;
    char buf[32];
    switch (t->token) {
        case STCSP_T_CONST:
            snprintf(buf, sizeof buf, "%d", t->num);
            out += buf;
            return;
        case STCSP_T_VAR: out += m.vars[t->var].name; return;
        case STCSP_T_ARR:
            out += m.array_names[t->arr] + "[";
            print_tree(m, t->right, out);
            out += "]";
            return;
        case STCSP_T_IF:
            out += "if (";
            print_tree(m, t->left, out);
            out += ") then (";
            print_tree(m, t->right->left, out);
            out += ") else (";
            print_tree(m, t->right->right, out);
            out += ")";
            return;
        case STCSP_T_ABS: case STCSP_T_NOT: case STCSP_T_FIRST: case STCSP_T_NEXT:
            out += token_text(t->token);
            out += "(";
            print_tree(m, t->right, out);
            out += ")";
            return;
        default: {
            bool root = is_constraint_root(t->token);
            if (!root) out += "(";
            print_tree(m, t->left, out);
            out += " ";
            out += token_text(t->token);
            out += " ";
            print_tree(m, t->right, out);
            if (!root) out += ")";
        }
    }
}

}  // namespace stcsp

// ------------------------------------------------------------------ C-ABI
using stcsp::Model;

struct stcsp_model {
    Model *m;
};

extern "C" {

int stcsp_model_load_text(const char *text, int prefix_k, stcsp_model **out) {
    if (!text || !out) return STCSP_E_INVALID;
    Model *m = nullptr;
    int rc = stcsp::build_model(text, prefix_k, &m);
    if (rc != STCSP_OK) return rc;
    *out = new stcsp_model{m};
    return STCSP_OK;
}

int stcsp_model_load_file(const char *path, int prefix_k, stcsp_model **out) {
    if (!path || !out) return STCSP_E_INVALID;
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        stcsp::set_error("cannot open %s", path);
        return STCSP_E_INVALID;
    }
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
    fclose(fp);
    return stcsp_model_load_text(text.c_str(), prefix_k, out);
}

const stcsp_problem *stcsp_model_problem(const stcsp_model *model) { return model ? &model->m->problem : nullptr; }

void stcsp_model_free(stcsp_model *model) {
    if (!model) return;
    delete model->m;
    delete model;
}

const char *stcsp_host_last_error(void) { return stcsp::g_last_error.c_str(); }

char *stcsp_model_constraint_string(const stcsp_model *model, int index) {
    if (!model || index < 0 || index >= (int)model->m->constraints.size()) return nullptr;
    std::string s;
    stcsp::print_tree(*model->m, model->m->constraints[index], s);
    return strdup(s.c_str());
}

void stcsp_host_free(void *p) { free(p); }

}  // extern "C"
