"""
ORACLE (test infrastructure only -- never imported by the product path).

Independent, deliberately simple (mailbox, pure Python) restatement of the parts of the reference's chess environment
that the hot path depends on:

  * board state + FEN in/out + UCI move application ... engine/src/environments/chess_related/board.cpp:117-275,
    boardstate.cpp:42-277 (on top of the un-vendored multi-variant Stockfish fork `github.com/QueensGambit/Stockfish`,
    pinned commit unknown -- engine/3rdparty/Stockfish is empty in the mount; its published rules are restated here and
    anchored on the reference's own tests: engine/tests/tests.cpp:158-1671)
  * legal move generation (chess, chess960, crazyhouse, 3check, KOTH) -- pseudo-legal + king-safety filter
  * board_to_planes, every layout except the legal-move dependent chess v2.7/2.8 ... inputrepresentation.cpp:33-680
  * policy labels, mirrored labels, policy-map index table, move -> index ... outputrepresentation.cpp:39-184,
    sfutil.cpp:142-285, DeepCrazyhouse/src/domain/variants/plane_policy_representation.py:22-224

Pinning: tests/test_oracle_chess.py checks this file against every plane golden of engine/tests/tests.cpp (sum / max /
key / argmax / FEN strings; transcribed with line numbers in tests/golden/planes_goldens.json), the rule tests
(castling, 3check, KOTH, crazyhouse drops, 3-fold), the frozen label list engine/tests/legacyconstants.h and the
FLAT_PLANE_IDX tables of policymaprepresentation.h (tests/golden/policy_tables.npz, made by oracle/make_golden_tables.py),
plus published perft counts.  Parity unpinned: legal-move SETS beyond those FENs/perft counts (the reference has no perft).
"""
from __future__ import annotations

import copy
from typing import List, Optional, Tuple

import numpy as np

FILES = "abcdefgh"
VARIANTS = {"chess": 0, "standard": 0, "chess960": 0, "fischerandom": 0, "crazyhouse": 1, "kingofthehill": 2, "3check": 3,
            "threecheck": 3, "antichess": 4, "atomic": 5, "horde": 6, "racingkings": 7}
START_FEN = {0: "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
             1: "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR[] w KQkq - 0 1",
             2: "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
             3: "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 3+3 0 1",
             4: "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w - - 0 1",
             5: "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
             6: "rnbqkbnr/pppppppp/8/1PP2PP1/PPPPPPPP/PPPPPPPP/PPPPPPPP/PPPPPPPP w kq - 0 1",
             7: "8/8/8/8/8/8/krbnNBRK/qrbnNBRQ w - - 0 1"}   # boardstate.h:322-385
MODE_CRAZYHOUSE, MODE_CHESS, MODE_LICHESS = 0, 1, 2
TERMINAL_LOSS, TERMINAL_DRAW, TERMINAL_WIN, TERMINAL_NONE = 0, 1, 2, 4

KNIGHT_D = [(1, 2), (2, 1), (2, -1), (1, -2), (-1, -2), (-2, -1), (-2, 1), (-1, 2)]
KING_D = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]
BISHOP_D = [(1, 1), (1, -1), (-1, -1), (-1, 1)]
ROOK_D = [(0, 1), (1, 0), (0, -1), (-1, 0)]


def sq(f, r):
    return r * 8 + f


def sq_name(s):
    return FILES[s & 7] + str((s >> 3) + 1)


def parse_sq(t):
    return (ord(t[1]) - 49) * 8 + (ord(t[0]) - 97)


class Board:
    """white pieces upper case, black lower case, None = empty; square a1 = 0 ... h8 = 63."""

    def __init__(self, fen: Optional[str] = None, is960: bool = False, variant: str = "chess"):
        self.variant = VARIANTS[variant]
        self.is960 = is960
        self.set(fen or START_FEN[self.variant])

    # ------------------------------------------------------------------------------------------------ setup / fen
    def set(self, fen: str):
        parts = fen.split()
        self.b: List[Optional[str]] = [None] * 64
        self.pocket = {c: 0 for c in "PNBRQpnbrq"}
        self.promoted = set()
        placement = parts[0]
        s, slashes, in_pocket, last = 56, 0, False, None
        for ch in placement:
            if ch == "[":
                in_pocket = True
            elif ch == "]":
                in_pocket = False
            elif in_pocket:
                if ch != "-":
                    self.pocket[ch] += 1
            elif ch == "/":
                slashes += 1
                if slashes == 8:
                    in_pocket = True
                else:
                    s -= 16
            elif ch.isdigit():
                s += int(ch)
            elif ch == "~":
                self.promoted.add(last)
            else:
                self.b[s] = ch
                last = s
                s += 1
        self.stm = 0 if parts[1] == "w" else 1
        # castling rights: letter -> rook square
        self.castle = {}
        self.castle_mask = {}
        for ch in parts[2] if len(parts) > 2 else "-":
            if ch == "-":
                continue
            white = ch.isupper()
            back = 0 if white else 56
            rook, king = ("R", "K") if white else ("r", "k")
            ks = [i for i in range(back, back + 8) if self.b[i] == king]
            if not ks:
                continue
            up = ch.upper()
            rs = None
            if up == "K":
                cand = [i for i in range(back + 7, back - 1, -1) if self.b[i] == rook]
                rs = cand[0] if cand else None
            elif up == "Q":
                cand = [i for i in range(back, back + 8) if self.b[i] == rook]
                rs = cand[0] if cand else None
            elif "A" <= up <= "H":
                rs = back + ord(up) - 65
                if self.b[rs] != rook:
                    rs = None
            if rs is not None:
                key = ("K" if rs > ks[0] else "Q") if white else ("k" if rs > ks[0] else "q")
                self.castle[key] = rs
                self.castle_mask.setdefault(ks[0], set()).add(key)
                self.castle_mask.setdefault(rs, set()).add(key)
        # en passant (kept only if a pawn of the side to move attacks it and the pushed pawn is behind it)
        self.ep = None
        if len(parts) > 3 and parts[3] != "-":
            e = parse_sq(parts[3])
            if self._ep_valid(e, self.stm):
                self.ep = e
        self.checks_given = [0, 0]
        nums = []
        for t in parts[4:]:
            if "+" in t:
                if t[0] == "+":
                    a, bb = t[1:].split("+")
                    self.checks_given = [int(a), int(bb)]
                else:
                    a, bb = t.split("+")
                    self.checks_given = [max(0, 3 - int(a)), max(0, 3 - int(bb))]
            else:
                nums.append(int(t))
        self.rule50 = nums[0] if nums else 0
        fullmove = nums[1] if len(nums) > 1 else 1
        self.ply = max(2 * (fullmove - 1), 0) + self.stm
        self.last_moves: List[Tuple[Optional[int], int]] = []   # (from or None for drops, to), most recent first
        self.history = [(self._key(), False)]                    # (key, that state's repetition != 0)
        self.repetition = 0

    def _ep_valid(self, e, stm):
        pawn_me, pawn_you = ("P", "p") if stm == 0 else ("p", "P")
        r = e >> 3
        if r != (5 if stm == 0 else 2) or self.b[e] is not None:
            return False
        behind = e - 8 if stm == 0 else e + 8
        if self.b[behind] != pawn_you:
            return False
        f = e & 7
        src_r = r - 1 if stm == 0 else r + 1
        return any(0 <= f + d < 8 and self.b[sq(f + d, src_r)] == pawn_me for d in (-1, 1))

    def _key(self):
        return (tuple(self.b), self.stm, tuple(sorted(self.castle.items())), self.ep,
                tuple(sorted(self.pocket.items())) if self.variant == 1 else None,
                tuple(self.checks_given) if self.variant == 3 else None)

    def fen(self) -> str:
        rows = []
        for r in range(7, -1, -1):
            row, empty = "", 0
            for f in range(8):
                p = self.b[sq(f, r)]
                if p is None:
                    empty += 1
                else:
                    if empty:
                        row += str(empty)
                        empty = 0
                    row += p + ("~" if self.variant == 1 and sq(f, r) in self.promoted else "")
            rows.append(row + (str(empty) if empty else ""))
        out = "/".join(rows)
        if self.variant == 1:
            out += "[" + "".join(c * self.pocket[c] for c in "QRBNPqrbnp") + "]"
        out += " w " if self.stm == 0 else " b "
        cs = ""
        for k in "KQkq":
            if k in self.castle:
                cs += (chr((65 if k.isupper() else 97) + (self.castle[k] & 7))) if self.is960 else k
        out += cs or "-"
        out += " " + (sq_name(self.ep) if self.ep is not None else "-")
        if self.variant == 3:
            out += f" {3 - self.checks_given[0]}+{3 - self.checks_given[1]}"
        out += f" {self.rule50} {1 + (self.ply - self.stm) // 2}"
        return out

    # ------------------------------------------------------------------------------------------------ attacks
    @staticmethod
    def _is_white(p):
        return p.isupper()

    def attacked_by(self, s, by_white, board=None):
        """list of squares holding pieces of the given colour that attack square s"""
        b = self.b if board is None else board
        f, r = s & 7, s >> 3
        out = []
        own = (lambda p: p is not None and p.isupper() == by_white)
        pr = r - 1 if by_white else r + 1   # rank a pawn of that colour must stand on
        for df in (-1, 1):
            if 0 <= f + df < 8 and 0 <= pr < 8:
                p = b[sq(f + df, pr)]
                if own(p) and p.upper() == "P":
                    out.append(sq(f + df, pr))
        for df, dr in KNIGHT_D:
            if 0 <= f + df < 8 and 0 <= r + dr < 8:
                p = b[sq(f + df, r + dr)]
                if own(p) and p.upper() == "N":
                    out.append(sq(f + df, r + dr))
        for df, dr in KING_D:
            if 0 <= f + df < 8 and 0 <= r + dr < 8:
                p = b[sq(f + df, r + dr)]
                if own(p) and p.upper() == "K":
                    out.append(sq(f + df, r + dr))
        for dirs, kinds in ((BISHOP_D, "BQ"), (ROOK_D, "RQ")):
            for df, dr in dirs:
                nf, nr = f + df, r + dr
                while 0 <= nf < 8 and 0 <= nr < 8:
                    p = b[sq(nf, nr)]
                    if p is not None:
                        if own(p) and p.upper() in kinds:
                            out.append(sq(nf, nr))
                        break
                    nf += df
                    nr += dr
        return out

    def king_sq(self, white, board=None):
        b = self.b if board is None else board
        k = "K" if white else "k"
        for i in range(64):
            if b[i] == k:
                return i
        return None

    def _kings_touch(self, board=None):
        a, b = self.king_sq(True, board), self.king_sq(False, board)
        return a is not None and b is not None and max(abs((a & 7) - (b & 7)), abs((a >> 3) - (b >> 3))) == 1

    def checkers(self):
        ks = self.king_sq(self.stm == 0)
        if ks is None or self.variant == 4:          # horde: White has no king; antichess: no check at all
            return []
        if self.variant == 5 and self._kings_touch():   # atomic: touching kings cannot be captured, hence never in check
            return []
        return self.attacked_by(ks, self.stm != 0)

    # ------------------------------------------------------------------------------------------------ move generation
    def _pseudo(self):
        """yields (uci, from, to, kind, extra); kind in normal|promo|ep|castle|drop.  Castling is king-takes-rook here."""
        white = self.stm == 0
        own = (lambda p: p is not None and p.isupper() == white)
        enemy = (lambda p: p is not None and p.isupper() != white)
        for s in range(64):
            p = self.b[s]
            if not own(p):
                continue
            f, r = s & 7, s >> 3
            t = p.upper()
            if t == "P":
                dr = 1 if white else -1
                promo_r = 7 if white else 0
                targets = []
                if 0 <= r + dr < 8 and self.b[sq(f, r + dr)] is None:
                    targets.append((sq(f, r + dr), "normal"))
                    two = r == (1 if white else 6) or (self.variant == 6 and white and r == 0)   # horde: first-rank pawns too
                    if two and self.b[sq(f, r + 2 * dr)] is None:
                        targets.append((sq(f, r + 2 * dr), "normal"))
                for df in (-1, 1):
                    if 0 <= f + df < 8 and 0 <= r + dr < 8:
                        d = sq(f + df, r + dr)
                        if enemy(self.b[d]):
                            targets.append((d, "normal"))
                        elif self.ep is not None and d == self.ep:
                            targets.append((d, "ep"))
                for d, kind in targets:
                    if (d >> 3) == promo_r:
                        for pc in ("qrbnk" if self.variant == 4 else "qrbn"):      # antichess: promotion to a king
                            yield (s, d, "promo", pc)
                    else:
                        yield (s, d, kind, None)
            elif t in "NK":
                for df, dr in (KNIGHT_D if t == "N" else KING_D):
                    if 0 <= f + df < 8 and 0 <= r + dr < 8 and not own(self.b[sq(f + df, r + dr)]):
                        yield (s, sq(f + df, r + dr), "normal", None)
            else:
                dirs = BISHOP_D if t == "B" else ROOK_D if t == "R" else BISHOP_D + ROOK_D
                for df, dr in dirs:
                    nf, nr = f + df, r + dr
                    while 0 <= nf < 8 and 0 <= nr < 8:
                        q = self.b[sq(nf, nr)]
                        if own(q):
                            break
                        yield (s, sq(nf, nr), "normal", None)
                        if q is not None:
                            break
                        nf += df
                        nr += dr
        # castling
        ks = self.king_sq(white)
        if ks is not None and self.variant != 4 and not self.attacked_by(ks, not white):      # no castling in antichess
            back = 0 if white else 56
            for key, oo in ((("K" if white else "k"), True), (("Q" if white else "q"), False)):
                if key not in self.castle:
                    continue
                rs = self.castle[key]
                kto, rto = back + (6 if oo else 2), back + (5 if oo else 3)
                need_empty = set(range(min(ks, kto), max(ks, kto) + 1)) | set(range(min(rs, rto), max(rs, rto) + 1))
                need_empty -= {ks, rs}
                if any(self.b[i] is not None for i in need_empty):
                    continue
                if any(self.attacked_by(i, not white) for i in range(min(ks, kto), max(ks, kto) + 1) if i != ks):
                    continue
                if self.is960:   # the rook must not be shielding the king's destination from a rook/queen on the back rank
                    b2 = list(self.b)
                    b2[rs] = None
                    if any(b2[a].upper() in "RQ" and ((a >> 3) == (kto >> 3) or (a & 7) == (kto & 7))
                           for a in self.attacked_by(kto, not white, b2)):
                        continue
                yield (ks, rs, "castle", oo)
        # drops
        if self.variant == 1:
            for pc in "PNBRQ":
                if self.pocket[pc if white else pc.lower()] == 0:
                    continue
                for d in range(64):
                    if self.b[d] is None and not (pc == "P" and (d >> 3) in (0, 7)):
                        yield (None, d, "drop", pc)

    def _apply(self, mv, board, pocket=None, promoted=None):
        """applies mv to a board copy (no bookkeeping); returns captured piece char (or None)"""
        frm, to, kind, extra = mv
        white = self.stm == 0
        cap = None
        if kind == "drop":
            board[to] = extra if white else extra.lower()
        elif kind == "castle":
            back = 0 if white else 56
            kto, rto = back + (6 if extra else 2), back + (5 if extra else 3)
            k, r = board[frm], board[to]
            board[frm] = None
            board[to] = None
            board[kto] = k
            board[rto] = r
        else:
            capsq = to
            if kind == "ep":
                capsq = to - 8 if white else to + 8
            cap = board[capsq]
            board[capsq] = None
            board[to] = board[frm]
            board[frm] = None
            if kind == "promo":
                board[to] = extra.upper() if white else extra.lower()
            self._blast = []
            if self.variant == 5 and cap is not None:      # atomic: the capturer and every non-pawn piece around `to` go as well
                board[to] = None
                f, r = to & 7, to >> 3
                for df, dr in KING_D:
                    if 0 <= f + df < 8 and 0 <= r + dr < 8:
                        d = sq(f + df, r + dr)
                        if board[d] is not None and board[d].upper() != "P":
                            board[d] = None
                            self._blast.append(d)
        return cap

    def legal_moves(self):
        white = self.stm == 0
        if self.variant == 4:                         # antichess: every pseudo-legal move, captures compulsory when there is one
            pseudo = list(self._pseudo())
            caps = [m for m in pseudo if m[2] == "ep" or self.b[m[1]] is not None]
            return caps if caps else pseudo
        out = []
        if self.variant == 5 and self.king_sq(white) is None:
            return list(self._pseudo())               # my king is already gone (the game is over): nothing left to protect
        for mv in self._pseudo():
            if self.variant == 5:                     # atomic
                if mv[2] != "castle" and self.b[mv[0]].upper() == "K" and self.b[mv[1]] is not None:
                    continue                          # kings never capture
                b2 = list(self.b)
                self._apply(mv, b2)
                ks, ko = self.king_sq(white, b2), self.king_sq(not white, b2)
                if ks is None:
                    continue                          # my own king would blow up
                if ko is None or self._kings_touch(b2) or not self.attacked_by(ks, not white, b2):
                    out.append(mv)                    # enemy king blown up (overrides check), touching kings, or simply safe
                continue
            b2 = list(self.b)
            self._apply(mv, b2)
            ks = self.king_sq(white, b2)
            if ks is not None and self.attacked_by(ks, not white, b2):
                continue
            if self.variant == 7:                     # racing kings: a move that gives check is illegal
                ok = self.king_sq(not white, b2)
                if ok is not None and self.attacked_by(ok, white, b2):
                    continue
            out.append(mv)
        return out

    def move_uci(self, mv) -> str:
        frm, to, kind, extra = mv
        if kind == "drop":
            return f"{extra}@{sq_name(to)}"
        if kind == "castle" and not self.is960:
            to = (frm >> 3) * 8 + (6 if extra else 2)
        return sq_name(frm) + sq_name(to) + (extra if kind == "promo" else "")

    def legal_uci(self):
        return sorted(self.move_uci(m) for m in self.legal_moves())

    def find_move(self, uci: str):
        for m in self.legal_moves():
            if self.move_uci(m) == uci:
                return m
        return None

    def parse_move_unchecked(self, uci: str):
        """UCI -> move tuple WITHOUT legality checking (for variants whose rules are not implemented, e.g. racing kings)."""
        if uci[1] == "@":
            return (None, parse_sq(uci[2:4]), "drop", uci[0].upper())
        frm, to = parse_sq(uci[0:2]), parse_sq(uci[2:4])
        p = self.b[frm]
        if len(uci) == 5:
            return (frm, to, "promo", uci[4].lower())
        if p.upper() == "P" and to == self.ep and (frm & 7) != (to & 7):
            return (frm, to, "ep", None)
        return (frm, to, "normal", None)

    # ------------------------------------------------------------------------------------------------ do move
    def push(self, mv):
        frm, to, kind, extra = mv
        white = self.stm == 0
        # Board::add_move_to_list (board.cpp:223-232): most recent first, capped at 8; castling keeps king/rook squares
        self.last_moves.insert(0, (None if kind == "drop" else frm, to))
        del self.last_moves[8:]
        self.ply += 1
        self.rule50 += 1
        moving = self.b[frm] if frm is not None else None
        new_ep = None
        if kind == "drop":
            self.pocket[extra if white else extra.lower()] -= 1
            self._apply(mv, self.b)
        elif kind == "castle":
            self._apply(mv, self.b)
        else:
            capsq = (to - 8 if white else to + 8) if kind == "ep" else to
            was_promoted = capsq in self.promoted
            cap = self._apply(mv, self.b)
            if cap is not None:
                self.rule50 = 0
                self.promoted.discard(capsq)
                if self.variant == 1:
                    t = "P" if was_promoted else cap.upper()
                    self.pocket[t if white else t.lower()] += 1
            if frm in self.promoted:
                self.promoted.discard(frm)
                self.promoted.add(to)
            if moving.upper() == "P":
                self.rule50 = 0
                if abs(to - frm) == 16 and not (self.variant == 6 and (frm >> 3) == (0 if white else 7)):   # horde: none from rank 1
                    mid = (to + frm) // 2
                    if self._ep_valid_after_push(mid, white):
                        new_ep = mid
                if kind == "promo" and self.variant == 1:
                    self.promoted.add(to)
        # castling rights: lost when something moves from / to the king's or that rook's original square (atomic: or is blown up)
        for s_ in (frm, to) + tuple(getattr(self, "_blast", ()) if kind not in ("drop", "castle") else ()):
            for key in self.castle_mask.get(s_, ()):
                self.castle.pop(key, None)
        self.ep = new_ep
        self.stm ^= 1
        if self.variant == 3 and self.checkers():
            self.checks_given[0 if white else 1] += 1
        # repetition (Stockfish: st->repetition = distance to previous occurrence, negative if that one was itself a repetition)
        key = self._key()
        n = len(self.history) + 1
        end = (n - 1) if self.variant == 1 else min(self.rule50, n - 1)
        self.repetition = 0
        i = 4
        while i <= end:
            k2, rep2 = self.history[n - 1 - i]
            if k2 == key:
                self.repetition = -i if rep2 else i
                break
            i += 2
        self.history.append((key, self.repetition != 0))

    def _ep_valid_after_push(self, mid, white_pushed):
        enemy_pawn = "p" if white_pushed else "P"
        f, r = mid & 7, mid >> 3
        src_r = r + 1 if white_pushed else r - 1   # rank where an enemy pawn attacking `mid` stands
        return any(0 <= f + d < 8 and self.b[sq(f + d, src_r)] == enemy_pawn for d in (-1, 1))

    def push_uci(self, uci: str, checked: bool = True):
        mv = self.find_move(uci) if checked else self.parse_move_unchecked(uci)
        if mv is None:
            raise ValueError(f"illegal move {uci} in {self.fen()}")
        self.push(mv)
        return mv

    def copy(self):
        return copy.deepcopy(self)

    # ------------------------------------------------------------------------------------------------ rules
    def number_repetitions(self):                         # board.cpp:132-141 (0 or 1 only)
        return 0 if self.repetition == 0 else 1

    def terminal(self):
        """BoardState::is_terminal (boardstate.cpp:143-226) for chess / crazyhouse / koth / 3check / antichess / horde / racing kings
        (the variant predicates are those of the multi-variant Stockfish fork the reference links: is_anti_win, is_horde_loss,
        is_race_win/draw/loss)."""
        n = len(self.legal_moves())
        me_white = self.stm == 0
        mine = [p for p in self.b if p is not None and p.isupper() == me_white]
        theirs = [p for p in self.b if p is not None and p.isupper() != me_white]
        if self.variant == 5:                         # is_atomic_win / is_atomic_loss
            if self.king_sq(not me_white) is None:
                return TERMINAL_WIN
            if self.king_sq(me_white) is None:
                return TERMINAL_LOSS
        if self.variant == 4:                         # is_anti_win / is_anti_loss
            if not mine:
                return TERMINAL_WIN
            if not theirs:
                return TERMINAL_LOSS
        if self.variant == 6:                         # is_horde_loss: the kingless side has nothing left and is to move
            horde_white = self.king_sq(True) is None
            if horde_white == me_white and not mine:
                return TERMINAL_LOSS
        if self.variant == 7:                         # is_race_win / is_race_draw / is_race_loss
            km, kt = self.king_sq(me_white), self.king_sq(not me_white)
            if km is not None and kt is not None:
                if (km >> 3) == 7:
                    return TERMINAL_DRAW if (kt >> 3) == 7 else TERMINAL_WIN
                if (kt >> 3) == 7:
                    if (km >> 3) < (7 if me_white else 6):
                        return TERMINAL_LOSS
                    can_follow = False
                    for df, dr in KING_D:
                        nf, nr = (km & 7) + df, (km >> 3) + dr
                        if nr == 7 and 0 <= nf < 8:
                            d = sq(nf, nr)
                            own_there = self.b[d] is not None and self.b[d].isupper() == me_white
                            if not own_there and not self.attacked_by(d, not me_white):
                                can_follow = True
                    if not can_follow:
                        return TERMINAL_LOSS
        if self.variant == 2:
            center = (27, 28, 35, 36)
            if self.king_sq(me_white) in center:
                return TERMINAL_WIN
            if self.king_sq(not me_white) in center:
                return TERMINAL_LOSS
        if self.variant == 3:
            if self.checks_given[self.stm] >= 3:
                return TERMINAL_WIN
            if self.checks_given[self.stm ^ 1] >= 3:
                return TERMINAL_LOSS
        if n == 0:
            if self.variant == 4:
                return TERMINAL_WIN                   # a stalemate is a win in antichess
            return TERMINAL_LOSS if self.checkers() else TERMINAL_DRAW
        if self.repetition < 0:
            return TERMINAL_DRAW
        if self.variant != 1 and self.rule50 > 99:
            return TERMINAL_DRAW
        if self.variant in (0, 5):
            pcs = [p for p in self.b if p is not None]
            nb = sum(p.upper() == "B" for p in pcs)
            nn = sum(p.upper() == "N" for p in pcs)
            if len(pcs) <= 4 and (len(pcs) == 2 or (len(pcs) == 3 and (nb == 1 or nn == 1)) or
                                  (len(pcs) == 4 and (pcs.count("N") == 2 or pcs.count("n") == 2))):
                return TERMINAL_DRAW
        return TERMINAL_NONE

    # ------------------------------------------------------------------------------------------------ game phase
    # Board::get_phase (board.cpp:540-587) with get_majors_and_minors_count (:446-449), is_backrank_sparse (:451-458), score_region
    # (:460-507) and get_mixedness (:509-538): the lichess Divider.  The phase column of the training samples
    # (TrainDataExporter::save_cur_phase, traindataexporter.cpp:91-103) and the exporter choice of self-play (selfplay.cpp:232-238).
    def majors_and_minors(self):
        return sum(1 for p in self.b if p is not None and p.upper() in "QRNB")

    def backrank_sparse(self):
        white = sum(1 for s in range(0, 8) if self.b[s] is not None and self.b[s].isupper())
        black = sum(1 for s in range(56, 64) if self.b[s] is not None and self.b[s].islower())
        return white <= 3 or black <= 3

    @staticmethod
    def _score_region(w, b, rank):                        # rank 1-based (board.cpp:460-507), branch by branch
        if (w, b) == (1, 0):
            return 1 + (8 - rank)
        if (w, b) == (2, 0):
            return 2 + (rank - 2 if rank > 2 else 0)
        if (w, b) in ((3, 0), (4, 0)):
            return 3 + (rank - 1 if rank > 1 else 0)
        if (w, b) == (0, 1):
            return 1 + rank
        if (w, b) == (1, 1):
            return 5 + abs(3 - rank)
        if (w, b) == (2, 1):
            return 4 + rank
        if (w, b) == (3, 1):
            return 5 + rank
        if (w, b) == (0, 2):
            return 2 + (6 - rank if rank < 6 else 0)
        if (w, b) == (1, 2):
            return 4 + (6 - rank)
        if (w, b) == (2, 2):
            return 7
        if (w, b) in ((0, 3), (0, 4)):
            return 3 + (7 - rank if rank < 7 else 0)
        if (w, b) == (1, 3):
            return 5 + (6 - rank)
        return 0

    def mixedness(self):
        mix = 0
        for r in range(7):
            for f in range(7):
                w = b = 0
                for dx in (0, 1):
                    for dy in (0, 1):
                        p = self.b[sq(f + dx, r + dy)]
                        if p is not None:
                            if p.isupper():
                                w += 1
                            else:
                                b += 1
                mix += self._score_region(w, b, r + 1)
        return mix

    def game_phase(self, num_phases: int, definition: int) -> int:
        if definition == 0:                               # LICHESS (the reference only asserts num_phases == 3)
            mm = self.majors_and_minors()
            if mm <= 6:
                return 2
            if mm <= 10 or self.backrank_sparse() or self.mixedness() > 150:
                return 1
            return 0
        if definition == 1:                               # MOVECOUNT
            if num_phases == 1:
                return 0
            phase_length = float(round(42.85 / num_phases))          # std::round: half away from zero; never a tie for small counts
            g = (self.ply // 2) / phase_length                      # total_move_cout() = gamePly / 2 (board.cpp:127-130)
            return num_phases - 1 if g > num_phases - 1 else int(g)
        return 0

    def perft(self, depth):
        moves = self.legal_moves()
        if depth <= 1:
            return len(moves)
        n = 0
        for m in moves:
            c = self.copy()
            c.push(m)
            n += c.perft(depth - 1)
        return n


# ======================================================================================================================
# input planes (inputrepresentation.cpp)
# ======================================================================================================================
LAYOUTS = {  # (mode, version major) -> name, channels
    (MODE_CRAZYHOUSE, 1): ("cz_v1", 34), (MODE_CRAZYHOUSE, 2): ("cz_v2", 51), (MODE_CRAZYHOUSE, 3): ("cz_v3", 64),
    (MODE_CHESS, 1): ("chess_v1", 39), (MODE_CHESS, 3): ("chess_v3", 52),
    (MODE_CHESS, "2.7"): ("chess_v27", 33), (MODE_CHESS, "2.8"): ("chess_v28", 38),     # make_version<2,7,0> / <2,8,0>
    (MODE_LICHESS, 2): ("lichess_v2", 63), (MODE_LICHESS, 3): ("lichess_v3", 80),
}


class _Planes:
    """PlaneData (inputrepresentation.cpp:48-109): a cursor over C x 64 floats, writing only non-zero entries."""

    def __init__(self, board: Board, channels: int, normalize: bool):
        self.bd = board
        self.x = np.zeros((channels, 64), np.float32)
        self.c = 0
        self.normalize = normalize
        self.flip = board.stm == 1 and board.variant != 7            # flip_board(), inputrepresentation.h:58-66
        self.me_white = board.stm == 0

    def fsq(self, s):
        return s ^ 56 if self.flip else s                            # vertical_flip, sfutil.h:135-137

    def plane_squares(self, squares):                                # set_plane_to_bitboard
        for s in squares:
            self.x[self.c, self.fsq(s)] = 1.0
        self.c += 1

    def plane_value(self, v, inc=True):
        self.x[self.c, :] = np.float32(v)
        if inc:
            self.c += 1

    def single(self, s, inc=True):
        self.x[self.c, self.fsq(s)] = 1.0
        if inc:
            self.c += 1

    # plane groups -----------------------------------------------------------------------------------------------
    def pieces(self):                                                # :112-122
        for white in (self.me_white, not self.me_white):
            for t in "PNBRQK":
                ch = t if white else t.lower()
                self.plane_squares([s for s in range(64) if self.bd.b[s] == ch])

    def repetition(self, rep):                                       # :124-136
        if rep >= 1:
            self.plane_value(1.0)
            if rep >= 2:
                self.plane_value(1.0)
                return
            self.c += 1
            return
        self.c += 2

    def pockets(self, max_prisoners):                                # :139-151
        for white in (self.me_white, not self.me_white):
            for t in "PNBRQ":
                cnt = self.bd.pocket[t if white else t.lower()]
                if cnt > 0:
                    self.plane_value(np.float32(cnt) / np.float32(max_prisoners) if self.normalize else cnt, inc=False)
                self.c += 1

    def promoted(self):                                              # :153-157
        for white in (self.me_white, not self.me_white):
            self.plane_squares([s for s in self.bd.promoted if self.bd.b[s] is not None and self.bd.b[s].isupper() == white])

    def ep(self):                                                    # :160-166
        if self.bd.ep is not None:
            self.single(self.bd.ep, inc=False)
        self.c += 1

    def color(self):                                                 # :168-175
        if self.me_white:
            self.plane_value(1.0)
        else:
            self.c += 1

    def total_moves(self):                                           # :177-181
        v = self.bd.ply // 2 + 1
        self.plane_value(np.float32(v) / np.float32(500) if self.normalize else v)

    def castling(self):                                              # :183-221
        order = "KQkq" if self.me_white else "kqKQ"
        for k in order:
            if k in self.bd.castle:
                self.plane_value(1.0, inc=False)
            self.c += 1

    def no_progress(self, max_np):                                   # :223-226
        self.plane_value(np.float32(self.bd.rule50) / np.float32(max_np) if self.normalize else self.bd.rule50)

    def remaining_checks(self):                                      # :229-247
        if self.bd.variant == 3:
            for col in (self.bd.stm, self.bd.stm ^ 1):
                g = self.bd.checks_given[col]
                if g != 0:
                    self.plane_value(1.0)
                    if g >= 2:
                        self.plane_value(1.0, inc=False)
                    self.c += 1
                else:
                    self.c += 2
            return
        self.c += 4

    def variant_and_960(self):                                       # :251-263
        if self.bd.is960:
            self.plane_value(1.0, inc=False)
        slot = {0: 1, 1: 2, 2: 3, 3: 4, 4: 5, 5: 6, 6: 7, 7: 8}[self.bd.variant]   # CHANNEL_MAPPING_VARIANTS
        self.x[self.c + slot, :] = 1.0
        self.c += 9

    def last_moves(self, nb_last_moves=8):                           # :266-282; NB_LAST_MOVES: 8, chess 2.x builds 1 (boardstate.h:171-176)
        pre = self.c
        for frm, to in self.bd.last_moves[:nb_last_moves]:
            if frm is None:
                self.c += 1
            else:
                self.single(frm)
            self.single(to)
        self.c = pre + 2 * nb_last_moves

    def check_moves(self, legal):                                    # :382-393: origins / destinations of the legal moves that give check
        for mv in legal:
            nxt = self.bd.copy()
            nxt.push(mv)
            if nxt.checkers():
                self.x[self.c, self.fsq(mv[0])] = 1.0                # castling: king square -> rook square (the fork's Move encoding)
                self.x[self.c + 1, self.fsq(mv[1])] = 1.0
        self.c += 2

    def mobility(self, legal):                                       # :395-398, NORMALIZE_MOBILITY 64 (boardstate.h:243-245)
        self.plane_value(np.float32(len(legal)) / np.float32(64) if self.normalize else len(legal))

    def is960(self):                                                 # :284-290
        if self.bd.is960:
            self.plane_value(1.0, inc=False)
        self.c += 1

    def piece_masks(self):                                           # :292-300
        for white in (self.me_white, not self.me_white):
            self.plane_squares([s for s in range(64) if self.bd.b[s] is not None and self.bd.b[s].isupper() == white])

    def checkerboard(self):                                          # :302-314 (never flipped)
        target = 1
        for row in range(8):
            for col in range(8):
                if col % 2 == target:
                    self.x[self.c, row * 8 + col] = 1.0
            target = 1 - target
        self.c += 1

    def _rel(self, v):                                               # :316-322
        if v != 0:
            self.plane_value(np.float32(v) / np.float32(8) if self.normalize else v, inc=False)
        self.c += 1

    def _cnt(self, white, t):
        return sum(1 for p in self.bd.b if p == (t if white else t.lower()))

    def material_diff(self, with_king=False):                        # :324-345
        for t in "PNBRQK" if with_king else "PNBRQ":
            self._rel(self._cnt(self.me_white, t) - self._cnt(not self.me_white, t))

    def opposite_bishops(self):                                      # :401-406
        wb = [s for s in range(64) if self.bd.b[s] == "B"]
        bb = [s for s in range(64) if self.bd.b[s] == "b"]
        if len(wb) == 1 and len(bb) == 1 and ((wb[0] & 7) + (wb[0] >> 3)) % 2 != ((bb[0] & 7) + (bb[0] >> 3)) % 2:
            self.plane_value(1.0, inc=False)
        self.c += 1

    def checkers(self):                                              # :376-379
        self.plane_squares(self.bd.checkers())

    def material_count(self, with_king=False):                       # :407-424
        for t in "PNBRQK" if with_king else "PNBRQ":
            self._rel(self._cnt(self.me_white, t))


def board_to_planes(board: Board, mode: int, version_major, normalize: bool, repetitions: Optional[int] = None):
    """-> float32 [C, 8, 8].  Dispatch of inputrepresentation.cpp:628-680.  version_major: the major number, or "2.7" / "2.8" (chess)."""
    if not isinstance(version_major, int):
        major, _, minor = str(version_major).partition(".")
        version_major = int(major)
        if mode == MODE_CHESS and version_major == 2:
            version_major = "2.8" if int(minor or 0) == 8 else "2.7"
    if mode == MODE_CRAZYHOUSE and version_major not in (2, 3):
        version_major = 1
    if mode == MODE_CHESS and version_major not in (3, "2.7", "2.8"):
        version_major = 1
    if mode == MODE_LICHESS and version_major != 3:
        version_major = 2
    name, C = LAYOUTS[(mode, version_major)]
    rep = board.number_repetitions() if repetitions is None else repetitions
    p = _Planes(board, C, normalize)
    if name in ("cz_v1", "cz_v2"):
        p.pieces(); p.repetition(rep); p.pockets(32); p.promoted(); p.ep(); p.color(); p.total_moves(); p.castling()
        p.no_progress(40)
        if name == "cz_v2":
            p.is960(); p.last_moves()
    elif name == "chess_v1":
        p.pieces(); p.repetition(rep); p.ep(); p.color(); p.total_moves(); p.castling(); p.no_progress(50); p.is960()
        p.last_moves()
    elif name in ("chess_v27", "chess_v28"):                      # :503-533
        legal = board.legal_moves()
        p.pieces(); p.ep(); p.castling(); p.last_moves(1); p.is960(); p.piece_masks(); p.checkerboard(); p.material_diff()
        p.opposite_bishops(); p.checkers(); p.check_moves(legal); p.mobility(legal)
        if name == "chess_v28":
            p.material_count()
    elif name in ("chess_v3", "cz_v3"):
        p.pieces(); p.repetition(rep); p.ep(); p.castling(); p.no_progress(40 if name == "cz_v3" else 50); p.last_moves()
        p.is960(); p.piece_masks(); p.checkerboard(); p.material_diff(); p.opposite_bishops(); p.checkers()
        p.material_count()
        if name == "cz_v3":
            p.pockets(32); p.promoted()
    else:  # lichess
        p.pieces(); p.repetition(rep); p.pockets(16); p.promoted(); p.ep()
        if name == "lichess_v3":
            p.c += 2
        else:
            p.color(); p.total_moves()
        p.castling(); p.no_progress(50); p.remaining_checks(); p.variant_and_960(); p.last_moves()
        if name == "lichess_v3":
            p.piece_masks(); p.checkerboard(); p.material_diff(True); p.opposite_bishops(); p.checkers()
            p.material_count(True)
    assert p.c == C, (name, p.c, C)
    return p.x.reshape(C, 8, 8)


def plane_statistics(x: np.ndarray):
    """get_stats_from_input_planes (engine/tests/tests.cpp:66-80): sum, max, argmax (first strict max), key = sum(i*x[i])."""
    flat = x.reshape(-1).astype(np.float32)
    s = float(np.sum(flat.astype(np.float64)))
    mx, arg = 0.0, 0
    for i, v in enumerate(flat):
        if v > mx:
            mx, arg = float(v), i
    key = float(np.sum(np.arange(flat.size, dtype=np.float64) * flat.astype(np.float64)))
    return s, mx, key, arg


# ======================================================================================================================
# policy (outputrepresentation.cpp, sfutil.cpp, plane_policy_representation.py)
# ======================================================================================================================
def generate_labels(mode: int) -> List[str]:
    labels = []
    promo = ["q", "r", "b", "n"] + (["k"] if mode == MODE_LICHESS else [])
    kfo = [-2, -1, -2, 1, 2, -1, 2, 1]
    kro = [-1, -2, 1, -2, -1, 2, 1, 2]
    for f in range(8):
        for r in range(8):
            dest = [(i, r) for i in range(8)] + [(f, i) for i in range(8)] + [(f + i, r + i) for i in range(-7, 8)] + \
                   [(f + i, r - i) for i in range(-7, 8)] + [(f + kfo[i], r + kro[i]) for i in range(8)]
            for f2, r2 in dest:
                if (f, r) != (f2, r2) and 0 <= f2 < 8 and 0 <= r2 < 8:
                    labels.append(FILES[f] + str(r + 1) + FILES[f2] + str(r2 + 1))
    for f in range(8):
        for p in promo:
            labels.append(FILES[f] + "2" + FILES[f] + "1" + p)
            labels.append(FILES[f] + "7" + FILES[f] + "8" + p)
            if f > 0:
                labels.append(FILES[f] + "2" + FILES[f - 1] + "1" + p)
                labels.append(FILES[f] + "7" + FILES[f - 1] + "8" + p)
            if f < 7:
                labels.append(FILES[f] + "2" + FILES[f + 1] + "1" + p)
                labels.append(FILES[f] + "7" + FILES[f + 1] + "8" + p)
    if mode != MODE_CHESS:
        for f in range(8):
            for r in range(8):
                for pc in "PNBRQ":
                    if pc != "P" or r not in (0, 7):
                        labels.append(pc + "@" + FILES[f] + str(r + 1))
    return labels


def mirror_label(l: str) -> str:                                     # sfutil.cpp:183-197
    return "".join(str(9 - int(ch)) if ch.isdigit() else ch for ch in l)


def flat_plane_index(label: str, mode: int) -> int:                  # plane_policy_representation.py:22-224
    pid = {"p": 0, "n": 1, "b": 2, "r": 3, "q": 4, "k": 5}
    if label[1] == "@":
        # Quirk kept for parity: the python generator moves drops to planes 79..83 once king promotions exist, but the
        # table the C++ engine ships for MODE_LICHESS (policymaprepresentation.h:2314-4631) still has them at 76..80,
        # i.e. overlapping the king-promotion planes 76..78 (e.g. "a2a1k" and "N@a2" both map to 77*64+8).
        return (76 + pid[label[0].lower()]) * 64 + parse_sq(label[2:4])
    frm, to = parse_sq(label[0:2]), parse_sq(label[2:4])
    dy, dx = (to >> 3) - (frm >> 3), (to & 7) - (frm & 7)
    if len(label) == 5:
        return (64 + (pid[label[4]] - 1) * 3 + dx + 1) * 64 + frm
    knight = [[2, 1], [1, 2], [-1, 2], [-2, 1], [-2, -1], [-1, -2], [1, -2], [2, -1]]
    if [dy, dx] in knight:
        return (56 + knight.index([dy, dx])) * 64 + frm
    length = max(abs(dx), abs(dy)) - 1
    cases = [dx == 0 and dy > 0, dx > 0 and dy > 0, dx > 0 and dy == 0, dy < 0 < dx, dx == 0 and dy < 0,
             dx < 0 and dy < 0, dx < 0 and dy == 0, dx < 0 < dy]
    return (cases.index(True) * 7 + length) * 64 + frm


class PolicyMap:
    def __init__(self, mode: int):
        self.mode = mode
        self.labels = generate_labels(mode)
        self.labels_mirrored = [mirror_label(l) for l in self.labels]
        self.flat = [flat_plane_index(l, mode) for l in self.labels]
        self.idx = {l: i for i, l in enumerate(self.labels)}
        self.idx_mirrored = {l: i for i, l in enumerate(self.labels_mirrored)}

    def index(self, board: Board, mv, is_policy_map: bool) -> int:
        """MV_LOOKUP / MV_LOOKUP_MIRRORED of the move's UCI string (node.cpp:961-979, boardstate.cpp:56-59)."""
        uci = board.move_uci(mv)
        mirrored = board.stm == 1 and board.variant != 7
        li = (self.idx_mirrored if mirrored else self.idx)[uci]
        return self.flat[li] if is_policy_map else li
