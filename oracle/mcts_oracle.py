"""
ORACLE (test infrastructure only -- never imported by the product path).

Python restatement of the reference's MCTS leaf-collection arithmetic on top of oracle.chess_oracle.Board:

  select_child_node / get_current_u_values / get_current_cput ... engine/src/node.cpp:1056-1063,1150-1167,1243-1246
  apply_virtual_loss_to_child / get_virtual_style ............ node.cpp:507-529, node.h:87-95
  revert_virtual_loss_and_update<> ........................... node.h:199-246
  revert_virtual_loss (collisions) ........................... node.cpp:655-679
  backup_value<> (tree case, two-player sign flip) ........... node.h:819-843
  create_mini_batch / get_new_child_to_evaluate .............. engine/src/searchthread.cpp:164-271,347-380
  fill_nn_results: prior gather, temperature, value .......... searchthread.cpp:290-299, node.cpp:956-979, util/blazeutil.h:77-87
  sort_moves_by_probabilities (stable, index tie-break) ...... node.cpp:464-470
  get_mcts_policy / first_and_second_max ..................... node.cpp:1070-1109, blazeutil.h:155-178
  solve_for_terminal, solved_win/loss/draw, end-in-ply ....... node.cpp:108-172, 268-297, 365-453

float32 / float64 mixing follows the C++ expression types (blaze vectors of float, uint32 and double); logf / powf are
taken from libm so that results are bit-identical to a C++ build on the same machine.

Parity status: the reference has NO unit test for this arithmetic ("parity unpinned") and its engine cannot be
compiled here (Stockfish fork + blaze are empty submodules), so this file is a line-by-line restatement checked by
hand-computed cases in tests/test_mcts.py, not by reference-generated vectors.  The MCTS solver (solve_for_terminal and friends,
node.cpp:108-172, 268-297, 365-453; solved branches of get_mcts_policy / get_best_action_index, node.cpp:299-363, 1123-1148) is
restated without tablebases.  useMCGS needs no restatement: the transposition link in Node::add_new_node_to_tree is unreachable
(node.cpp:730-731 takes the candidate from the child slot that is still empty at that point), so NODE_TRANSPOSITION is never
produced and "mcgs" searches the same tree as "mcts".  The epsilon exploration (searchthread.cpp:124-185, 451-473, 497-501) is restated with the reference's
rand() replaced by a seeded ANSI-C LCG (the reference seeds rand() with the time, so its own runs do not replay either).
Dirichlet noise at the root (mctsagent.cpp:311-316, node.cpp:950-954, blazeutil.h:113-124) draws its gamma variates from the
C++ standard library through oracle/stdgamma.cpp, seeded instead of std::random_device.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math

import numpy as np

from . import chess_oracle as co

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]

F = np.float32
Q_INIT = F(-1.0)
VIRTUAL_LOSS, VIRTUAL_VISIT, VIRTUAL_OFFSET, VIRTUAL_MIX = 0, 1, 2, 3
NT_WIN, NT_DRAW, NT_LOSS, NT_UNSOLVED = 0, 1, 2, 6      # NodeType, nodedata.h:42-52 (tablebase build numbering)


class Settings:
    def __init__(self, **kw):
        self.batch_size = 16
        self.cpuct_init = F(2.5)
        self.cpuct_base = F(19652.0)
        self.node_policy_temperature = F(1.7)
        self.virtual_style = VIRTUAL_MIX
        self.virtual_mix_threshold = 1000
        self.virtual_offset_strength = 0.001
        self.q_value_weight = F(1.0)
        self.q_veto_delta = F(0.4)
        self.mode = co.MODE_CRAZYHOUSE
        self.is_policy_map = True
        self.epsilon_greedy_counter = 0      # 1 / Centi_Epsilon_Greedy: UCI default 20 (optionsuci.cpp:90, crazyara.cpp:749); 0 = off
        self.epsilon_checks_counter = 0      # UCI default 100
        self.seed = 1
        self.mcts_solver = True              # MCTS_Solver, optionsuci.cpp:129
        self.dirichlet_epsilon = F(0.0)      # Centi_Dirichlet_Epsilon: 0 (25 in RL builds), optionsuci.cpp:84-86
        self.dirichlet_alpha = F(0.2)        # Centi_Dirichlet_Alpha 20
        for k, v in kw.items():
            setattr(self, k, v)


def get_current_cput(visits, s):
    v = F(visits)
    return F(_libm.logf(F(F(v + s.cpuct_base + F(1)) / s.cpuct_base))) + s.cpuct_init


def virtual_style(s, visits):
    if s.virtual_style == VIRTUAL_MIX:
        return VIRTUAL_LOSS if visits > s.virtual_mix_threshold else VIRTUAL_VISIT
    return s.virtual_style


class Node:
    def __init__(self, board: co.Board, policy_map: co.PolicyMap, s: Settings):
        self.moves = board.legal_moves()
        self.terminal = False
        self.value_sum = 0.0
        self.real_visits = 0
        self.has_nn = False
        self.sorted = False
        self.has_data = False
        self.inspected = False
        self.visit_sum = 0
        self.free_visits = 0
        self.no_visit_idx = 0
        self.node_type = NT_UNSOLVED
        self.end_in_ply = 0
        self.checkmate_idx = -1
        self.unsolved_children = 0
        t = board.terminal()
        if t != co.TERMINAL_NONE:
            self.terminal = True
            self.has_data = True
            self.sorted = True
            self.set_value({co.TERMINAL_WIN: F(1), co.TERMINAL_DRAW: F(0), co.TERMINAL_LOSS: F(-1)}[t])
            self.node_type = {co.TERMINAL_WIN: NT_WIN, co.TERMINAL_DRAW: NT_DRAW, co.TERMINAL_LOSS: NT_LOSS}[t]
            if t == co.TERMINAL_DRAW:
                self.moves = []
        self.uci = [board.move_uci(m) for m in self.moves]
        self.priors = [F(0)] * len(self.moves)
        self.policy_idx = [] if self.terminal else [policy_map.index(board, m, s.is_policy_map) for m in self.moves]
        self.child_visits, self.q, self.child, self.vl, self.child_types = [], [], [], [], []

    def set_value(self, v):
        self.real_visits += 1
        self.value_sum = float(F(v) * F(self.real_visits))

    def value(self):
        return F(self.value_sum / self.real_visits)

    def real_child_visits(self, c):
        return self.child_visits[c] - self.vl[c]


def first_and_second_max(values):
    """first_and_second_max (blazeutil.h:155-178): (max, runner-up, argmax, arg of the runner-up); the runner-up starts at
    numeric_limits<T>::min(), the smallest POSITIVE number.  Pinned by the reference's own test (engine/tests/tests.cpp:626-646)."""
    first, second, fa, sa = values[0], 2.2250738585072014e-308, 0, 0
    for i in range(1, len(values)):
        if values[i] > first:
            second, sa, first, fa = first, fa, values[i], i
        elif values[i] > second:
            second, sa = values[i], i
    return first, second, fa, sa


class Tree:
    def __init__(self, board: co.Board, s: Settings, clone_keeps_last_moves=None):
        self.s = s
        self.root_board = board
        self.pm = co.PolicyMap(s.mode)
        self.keep = (s.mode != co.MODE_CRAZYHOUSE) if clone_keeps_last_moves is None else clone_keeps_last_moves
        self.root = Node(board, self.pm, s)
        self.new_nodes, self.new_traj, self.coll_traj = [], [], []
        self.rng = s.seed & 0xFFFFFFFF
        self.noise_engine = None             # std::default_random_engine, created on first use (oracle/stdgamma.cpp)

    # ---------------------------------------------------------------------------------------------------------------
    def fill_nn_result(self, n: Node, value, probs):
        n.priors = [F(probs[i]) for i in n.policy_idx]
        n.policy_idx = []
        t = self.s.node_policy_temperature
        if t != 1:
            e = F(F(1.0) / F(t))
            n.priors = [F(_libm.powf(p, e)) for p in n.priors]
            tot = F(0)
            for p in n.priors:
                tot = F(tot + p)
            n.priors = [F(p / tot) for p in n.priors]
        n.set_value(F(value))
        n.has_nn = True

    def set_root_result(self, value, probs):
        self.fill_nn_result(self.root, value, probs)
        self.prepare(self.root)

    def begin_search(self):
        """Start of a `go` (mctsagent.cpp:311-316): apply_dirichlet_noise_to_prior_policy (node.cpp:950-954) with
        get_dirichlet_noise (blazeutil.h:113-124), then fully_expand_node (node.cpp:582-593)."""
        n, s = self.root, self.s
        if not (F(s.dirichlet_epsilon) > F(0.009)) or n.terminal or not n.has_nn or not n.moves:
            return
        lib = _stdgamma()
        if self.noise_engine is None:
            self.noise_engine = lib.stdgamma_new(ctypes.c_uint(s.seed & 0xFFFFFFFF))
        buf = (ctypes.c_float * len(n.moves))()
        lib.stdgamma_draw(self.noise_engine, ctypes.c_float(float(s.dirichlet_alpha)), len(n.moves), buf)
        noise = [F(v) for v in buf]
        tot = F(0)
        for v in noise:
            tot = F(tot + v)
        keep = F(F(1) - F(s.dirichlet_epsilon))
        n.priors = [F(F(keep * p) + F(F(s.dirichlet_epsilon) * F(v / tot))) for p, v in zip(n.priors, noise)]
        if not n.sorted:
            self.prepare(n)
        while n.no_visit_idx < len(n.moves):
            self.increment_no_visit_idx(n)

    def apply_move(self, uci):
        """MCTSAgent::apply_move_to_tree + get_root_node_from_tree (mctsagent.cpp:130-164,230-247): the subtree below the played
        move becomes the tree if that child is a playout node with visits; otherwise the tree restarts at the new position."""
        r = self.root
        move = None
        for m in (r.moves if r.moves else self.root_board.legal_moves()):
            if self.root_board.move_uci(m) == uci:
                move = m
        if move is None:
            for m in self.root_board.legal_moves():
                if self.root_board.move_uci(m) == uci:
                    move = m
        assert move is not None, "illegal move " + uci
        child = None
        if r.has_data:
            for i in range(r.no_visit_idx):
                if r.uci[i] == uci:
                    child = r.child[i]
        self.root_board = self.root_board.copy()
        self.root_board.push(move)
        keep = child is not None and child.has_data and child.has_nn and not child.terminal and child.visit_sum > 0
        self.root = child if keep else Node(self.root_board, self.pm, self.s)
        return keep

    def prepare(self, n: Node):
        order = sorted(range(len(n.moves)), key=lambda i: (-float(n.priors[i]), i))
        n.moves = [n.moves[i] for i in order]
        n.uci = [n.uci[i] for i in order]
        n.priors = [n.priors[i] for i in order]
        n.sorted = True
        if not n.has_data:
            n.has_data = True
            n.no_visit_idx = 1
            n.child_visits, n.q, n.child, n.vl, n.child_types = [0], [Q_INIT], [None], [0], [NT_UNSOLVED]
            n.unsolved_children = len(n.moves)

    def increment_no_visit_idx(self, n: Node):
        if n.no_visit_idx < len(n.moves):
            n.no_visit_idx += 1
            n.child_visits.append(0)
            n.q.append(Q_INIT)
            n.child.append(None)
            n.vl.append(0)
            n.child_types.append(NT_UNSOLVED)

    def select_child(self, n: Node):
        if not n.sorted:
            self.prepare(n)
        if n.no_visit_idx == 1:
            return 0
        if n.checkmate_idx >= 0:             # has_forced_win
            return n.checkmate_idx
        cpuct = get_current_cput(n.visit_sum, self.s)
        sq = math.sqrt(float(n.visit_sum))
        best, best_v = 0, F(-np.inf)
        for i in range(n.no_visit_idx):
            u = F(float(F(cpuct * n.priors[i])) * (sq / (float(n.child_visits[i]) + 1.0)))
            v = F(n.q[i] + u)
            if v > best_v:
                best, best_v = i, v
        return best

    def apply_virtual_loss(self, n: Node, c):
        st = virtual_style(self.s, n.child_visits[c])
        if st == VIRTUAL_LOSS:
            n.q[c] = F((float(n.q[c]) * n.child_visits[c] - 1) / float(n.child_visits[c] + 1))
        elif st == VIRTUAL_OFFSET:
            n.q[c] = F(float(n.q[c]) - self.s.virtual_offset_strength)
        n.child_visits[c] += 1
        n.visit_sum += 1
        n.vl[c] += 1

    def revert_virtual_loss(self, n: Node, c):
        st = virtual_style(self.s, n.child_visits[c])
        if st == VIRTUAL_LOSS:
            n.q[c] = F((float(n.q[c]) * n.child_visits[c] + 1) / (n.child_visits[c] - 1))
        elif st == VIRTUAL_OFFSET:
            n.q[c] = F(float(n.q[c]) + self.s.virtual_offset_strength)
        n.child_visits[c] -= 1
        n.visit_sum -= 1
        n.vl[c] -= 1

    def revert_virtual_loss_and_update(self, n: Node, c, value, free_backup, solve=False):
        value = F(value)
        n.value_sum += float(value)
        n.real_visits += 1
        if n.child_visits[c] == 1:
            n.q[c] = value
        else:
            st = virtual_style(self.s, n.child_visits[c])
            if st == VIRTUAL_LOSS:
                n.q[c] = F((float(n.q[c]) * n.child_visits[c] + 1 + float(value)) / n.child_visits[c])
            elif st == VIRTUAL_VISIT:
                r = n.real_child_visits(c)
                n.q[c] = F((float(n.q[c]) * r + float(value)) / (r + 1))
            elif st == VIRTUAL_OFFSET:
                r = n.real_child_visits(c)
                nq = float(n.q[c]) + n.vl[c] * self.s.virtual_offset_strength
                nq = (nq * r + float(value)) / (r + 1.0)
                n.q[c] = F(nq - ((n.vl[c] - 1) * self.s.virtual_offset_strength))
        n.vl[c] -= 1
        if free_backup:
            n.free_visits += 1
        if solve:
            self.solve_for_terminal(n, c)

    def solve_for_terminal(self, n: Node, c):
        """Node::solve_for_terminal (node.cpp:365-453), two-player, no tablebases."""
        ch = n.child[c]
        if ch is None or not ch.has_data:
            return False
        if ch.node_type == NT_UNSOLVED:
            return False
        if n.node_type != NT_UNSOLVED:
            return False
        if n.child_types[c] == NT_UNSOLVED:
            n.unsolved_children -= 1
            n.child_types[c] = ch.node_type
            if ch.node_type == NT_WIN:                       # disable_action (node.cpp:1006-1010)
                n.priors[c] = F(0)
                n.q[c] = F(-2147483647)
        if ch.node_type == NT_LOSS:                          # solved_win
            n.node_type = NT_WIN
            n.end_in_ply = ch.end_in_ply + 1                 # define_end_ply_for_solved_terminal, WIN branch
            n.set_value(F(1))                                # update_solved_terminal
            n.q[c] = F(1)
            n.checkmate_idx = c
            return True
        if n.unsolved_children == 0 and ch.node_type == NT_WIN and all(k.node_type == NT_WIN for k in n.child):   # solved_loss
            n.node_type = NT_LOSS
            for k in n.child:                                # longest line
                if k.end_in_ply + 1 > n.end_in_ply:
                    n.end_in_ply = k.end_in_ply + 1
            n.set_value(F(-1))
            n.q[c] = F(-1)
            return True
        if n.unsolved_children == 0 and ch.node_type != NT_LOSS:                                                  # solved_draw
            if all(k.has_data and k.node_type in (NT_DRAW, NT_WIN) for k in n.child) and any(k.node_type == NT_DRAW for k in n.child):
                n.node_type = NT_DRAW
                for k in n.child:                            # "shortest" line: never below the initial 0 (node.cpp:277-285)
                    if k.node_type == NT_DRAW and k.end_in_ply + 1 < n.end_in_ply:
                        n.end_in_ply = k.end_in_ply + 1
                n.set_value(F(0))
                n.q[c] = F(0)
                return True
        return False

    def backup_value(self, value, traj, free_backup, solve=False):
        value = F(value)
        for n, c in reversed(traj):
            value = F(-value)
            self.revert_virtual_loss_and_update(n, c, value, free_backup, solve)

    def best_action_index_fast(self, n: Node):
        """get_best_action_index(fast=True), node.cpp:1123-1148"""
        if n.checkmate_idx >= 0:
            return n.checkmate_idx
        best = 0
        if n.node_type == NT_LOSS:
            longest = 0
            for i, k in enumerate(n.child):
                if k.end_in_ply > longest:
                    longest, best = k.end_in_ply, i
            return best
        for i in range(1, n.no_visit_idx):
            if n.child_visits[i] > n.child_visits[best]:
                best = i
        return best

    # ---------------------------------------------------------------------------------------------------------------
    # ---- epsilon exploration --------------------------------------------------------------------------------------
    def next_rand(self):
        """rand(): classic ANSI-C LCG, 15-bit output."""
        self.rng = (self.rng * 1103515245 + 12345) & 0xFFFFFFFF
        return (self.rng >> 16) & 0x7FFF

    def get_random_depth(self):
        """searchthread.cpp:497-501: ceil(-log2(1 - r/100) - 1), r uniform in 1..100."""
        r = self.next_rand() % 100 + 1
        if r == 100:
            return 0          # size_t(+inf) in the reference: undefined behaviour, 0 with the x86-64 code GCC emits (checked on oracle/_ref)
        return int(math.ceil(-math.log2(1 - r / 100.0) - 1))

    def get_starting_node(self, cur, board):
        """searchthread.cpp:144-162: follow the most-visited line for a random number of plies (no virtual loss, no trajectory)."""
        child_idx, depth = -1, 0
        for _ in range(self.get_random_depth()):
            best = self.best_action_index_fast(cur)
            child_idx = best
            nxt = cur.child[best] if cur.no_visit_idx else None
            if nxt is None or not nxt.has_data or nxt.visit_sum < self.s.epsilon_greedy_counter or nxt.node_type != NT_UNSOLVED:
                break
            board.push(cur.moves[best])
            cur = nxt
            depth += 1
        return cur, child_idx

    def random_playout(self, cur):
        """searchthread.cpp:124-142"""
        if cur.no_visit_idx == len(cur.moves):
            idx = self.next_rand() % len(cur.moves)
            child = cur.child[idx]
            if child is None or not child.has_data:
                return idx
            if child.node_type == NT_UNSOLVED:
                return idx
            return -1
        idx = min(cur.no_visit_idx, len(cur.moves) - 1)
        self.increment_no_visit_idx(cur)
        return idx

    def select_enhanced_move(self, cur, board):
        """searchthread.cpp:451-473: make sure a checking move has been tried once."""
        if cur.has_data and not cur.inspected and not cur.terminal:
            first = cur.no_visit_idx
            for ci in range(first, len(cur.moves)):
                b2 = board.copy()
                b2.push(cur.moves[ci])
                if b2.checkers():
                    for _ in range(first, ci + 1):
                        self.increment_no_visit_idx(cur)
                    return ci
            cur.inspected = True
        return -1

    def get_new_child(self):
        cur = self.root
        board = self.root_board.copy()
        if not self.keep:
            board.last_moves = []
        traj = []
        forced = -1
        s = self.s
        if s.epsilon_greedy_counter and self.root.has_data and self.next_rand() % s.epsilon_greedy_counter == 0:
            cur, forced = self.get_starting_node(cur, board)
            forced = self.random_playout(cur)
        elif s.epsilon_checks_counter and self.root.has_data and self.next_rand() % s.epsilon_checks_counter == 0:
            cur, forced = self.get_starting_node(cur, board)
            forced = self.select_enhanced_move(cur, board)
            if forced < 0:
                forced = self.random_playout(cur)
        while True:
            c = forced if forced >= 0 else self.select_child(cur)
            forced = -1
            self.apply_virtual_loss(cur, c)
            traj.append((cur, c))
            nxt = cur.child[c]
            if nxt is None:
                board.push(cur.moves[c])
                self.increment_no_visit_idx(cur)
                nn = Node(board, self.pm, self.s)
                cur.child[c] = nn
                if nn.terminal:
                    return "terminal", nn, traj, None
                return "new", nn, traj, board
            if nxt.terminal:
                return "terminal", nxt, traj, None
            if not nxt.has_nn:
                return "collision", nxt, traj, None
            board.push(cur.moves[c])
            cur = nxt

    def collect(self, quota):
        """-> list of boards to evaluate (one per new leaf)"""
        boards = []
        n_term = 0
        if self.root.terminal or not self.root.has_nn or self.root.node_type != NT_UNSOLVED:
            return boards
        while len(boards) < quota and len(self.coll_traj) != quota and n_term < 2 * max(quota, 1):
            kind, node, traj, board = self.get_new_child()
            if kind == "terminal":
                n_term += 1
                self.backup_value(node.value(), traj, True, self.s.mcts_solver)
            elif kind == "collision":
                self.coll_traj.append(traj)
            else:
                self.new_nodes.append(node)
                self.new_traj.append(traj)
                boards.append(board)
        return boards

    def finish_batch(self, values, probs):
        for i, n in enumerate(self.new_nodes):
            self.fill_nn_result(n, values[i], probs[i])
        for n, traj in zip(self.new_nodes, self.new_traj):
            self.backup_value(n.value(), traj, False)
        self.new_nodes, self.new_traj = [], []
        for traj in self.coll_traj:
            for n, c in reversed(traj):
                self.revert_virtual_loss(n, c)
        self.coll_traj = []

    def node_count(self):
        return self.root.visit_sum - self.root.free_visits

    def best_move(self):
        n = self.root
        m = n.no_visit_idx

        def finish(pol):
            tot = sum(pol)
            with np.errstate(all="ignore"):
                pol = [float(np.float64(p) / np.float64(tot)) for p in pol]
            best = 0
            for i in range(m):
                if pol[i] > pol[best]:
                    best = i
            return n.uci[best], pol

        if n.node_type == NT_WIN:            # mcts_policy_based_on_wins, node.cpp:299-322
            return finish([1.0 if (k is not None and k.has_data and k.node_type == NT_LOSS) else 0.0 for k in n.child[:m]])
        if n.node_type == NT_LOSS:           # mcts_policy_based_on_losses, node.cpp:324-341
            pol, longest, li = [0.0] * m, 0, 0
            for i, k in enumerate(n.child[:m]):
                if k is not None and k.has_data and k.end_in_ply > longest:
                    longest, li = k.end_in_ply, i
            pol[li] = 1.0
            return finish(pol)
        pol = [float(v) for v in n.child_visits[:m]]
        if n.unsolved_children != len(n.moves):          # prune_losses_in_mcts_policy, node.cpp:343-363
            for i, k in enumerate(n.child[:m]):
                if k is not None and k.has_data and k.node_type == NT_WIN:
                    pol[i] = 0.0
        best_q = 0
        for i in range(1, m):
            if n.q[i] > n.q[best_q]:
                best_q = i
        first, second, fa, sa = first_and_second_max(pol)
        if self.s.q_value_weight > 0:
            if self.s.q_veto_delta != 0 and best_q != fa and n.q[best_q] > F(n.q[fa] + self.s.q_veto_delta) and n.child_visits[best_q] > 1:
                if pol[fa] > pol[best_q]:
                    pol[best_q], pol[fa] = pol[fa], pol[best_q]
            elif fa != sa and n.q[sa] > n.q[fa]:
                q_diff = F(n.q[sa] - n.q[fa])
                pol[sa] += float(F(q_diff * self.s.q_value_weight)) * pol[fa]
        return finish(pol)


_STDGAMMA = None


def _stdgamma():
    """ctypes handle on oracle/_build/libstdgamma.so (the C++ standard library's gamma_distribution, see stdgamma.cpp)."""
    global _STDGAMMA
    if _STDGAMMA is None:
        from . import build_oracle
        lib = ctypes.CDLL(build_oracle.build())
        lib.stdgamma_new.restype = ctypes.c_void_p
        lib.stdgamma_new.argtypes = [ctypes.c_uint]
        lib.stdgamma_draw.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        lib.stdgamma_draw.restype = None
        _STDGAMMA = lib
    return _STDGAMMA


def run_search(tree: Tree, evaluate, simulations, quota):
    """evaluate(list of Boards) -> (values, probs).  Mirrors SearchPool::run for a single tree / single lane."""
    if not tree.root.has_nn and not tree.root.terminal:
        v, p = evaluate([tree.root_board])
        tree.set_root_result(v[0], p[0])
    if len(tree.root.moves) == 1 and not tree.root.terminal:
        # MCTSAgent::evaluate_board_state (mctsagent.cpp:303-307): "Only single move available -> early stopping" -- no search.
        # handle_single_move's value hand-over from the previous search is agent state the tests of this oracle do not reach;
        # the reference build (oracle/_ref, tests/test_mcts_reference_build.py) covers it.
        return
    tree.begin_search()
    # nodes_limits_ok (searchthread.cpp:326-331): the limit is absolute on the root's visit counter (reused visits count)
    while not tree.root.terminal and tree.root.node_type == NT_UNSOLVED and tree.root.visit_sum < simulations:
        boards = tree.collect(quota)
        if boards:
            v, p = evaluate(boards)
            tree.finish_batch(v, p)
        else:
            tree.finish_batch([], [])
