"""Self-play / arena game loop on top of the search pool.

Follows engine/src/rl/selfplay.cpp (`generate_game` :192-265, `generate_arena_game` :267-308, `go_arena` :387-424,
`init_starting_state_from_raw_policy` :426-452, `play_move_and_update` :38-54), agents/agent.cpp (`set_best_move` :38-55) and
rl/gamepgn.cpp (:28-56) -- restructured for the many-trees pool: G games are played CONCURRENTLY, every game owns one tree
slot, one `run` of the pool searches the next move of all of them in shared GPU batches, then every game picks and plays its
move (the searched subtree is kept: mi_search_apply_move) and finished games are replaced by new ones.

Where the reference draws from rand() / std::random_device (opening plies, temperature sampling, resignation, node-count
jitter) this loop draws from one seeded numpy generator per game, so a run replays; the distributions are the reference's.
Training-sample export (traindataexporter.cpp) is not part of this loop.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import env, search

RESULT_STR = {1: "1-0", -1: "0-1", 0: "1/2-1/2"}          # from White's point of view (result[] in gamepgn / constants)


@dataclass
class SelfPlaySettings:
    variant: str = "crazyhouse"
    is960: bool = False
    simulations: int = 800              # search budget per move (SearchLimits::simulations; `nodes` works the same way)
    nodes: int = 0
    node_random_factor: float = 0.0     # RLSettings::nodeRandomFactor: +- factor/2 jitter of the node budget (selfplay.cpp:146-152)
    mean_init_ply: float = 0.0          # PlaySettings::meanInitPly: plies sampled from the raw policy ~ round(Exp(mean)) (:196-197)
    max_init_ply: int = 30              # PlaySettings::maxInitPly
    raw_policy_prob_temperature: float = 0.0   # RLSettings::rawPolicyProbabilityTemperature (apply_raw_policy_temp, :474-488)
    init_temperature: float = 0.0       # PlaySettings::initTemperature: sample the move from the MCTS policy while
    temperature_moves: int = 0          #   ply < temperatureMoves (agent.cpp:38-55)
    temperature_decay: float = 1.0      # PlaySettings::temperatureDecayFactor (playsettings.cpp:31-34)
    quantile_clipping: float = 0.0      # PlaySettings::quantileClipping
    resign_probability: float = 0.0     # RLSettings::resignProbability (:162-167)
    resign_threshold: float = -0.9      # RLSettings::resignThreshold on bestMoveQ (:169-182)
    reuse_tree: bool = True             # RLSettings::reuseTreeForSelpay
    max_plies: int = 600                # safety net: adjudicated as a draw
    seed: int = 1
    event: str = "SelfPlay"
    white: str = "crazyara-amd"
    black: str = "crazyara-amd"


@dataclass
class GameRecord:
    start_fen: str
    variant: str
    san: List[str] = field(default_factory=list)
    uci: List[str] = field(default_factory=list)
    book_plies: int = 0
    result: Optional[int] = None        # +1 white win, -1 black win, 0 draw
    termination: str = ""
    white: str = ""
    black: str = ""
    event: str = ""

    def pgn(self) -> str:
        """GamePGN's operator<< (gamepgn.cpp:28-56)."""
        res = RESULT_STR[self.result] if self.result is not None else "*"
        head = [f'[Variant "{self.variant}"]', f'[Event "{self.event}"]', f'[Date "{time.strftime("%Y.%m.%d %X")}"]',
                '[Site "MI355X"]', '[Round "?"]', f'[FEN "{self.start_fen}"]', f'[White "{self.white}"]',
                f'[Black "{self.black}"]', f'[Result "{res}"]', f'[PlyCount "{len(self.san)}"]', '[TimeControl "-"]', ""]
        body = []
        for ply, mv in enumerate(self.san):
            if ply % 2 == 0:
                body.append(f"{ply // 2 + 1}. ")
            body.append(mv + " ")
            if (ply + 1) % 8 == 0:
                body.append("\n")
        return "\n".join(head) + "\n" + "".join(body) + res + "\n\n"


def apply_temperature(p: np.ndarray, t: float) -> np.ndarray:
    """blazeutil.h:77-87: p^(1/T) renormalised; T == 1 leaves p untouched."""
    if t == 1:
        return p
    q = np.power(p, 1.0 / t)
    return q / q.sum()


def get_quantile(p: np.ndarray, quantile: float) -> float:
    """get_quantile (blazeutil.h:188-212), literally: sort ascending; 0 if the smallest entry already reaches the quantile; else
    accumulate (in float32) from the SECOND smallest entry and return the entry BEFORE the one that crosses the quantile, plus
    FLT_EPSILON.  Checked against the compiled reference function in tests/test_mcts_reference_build.py."""
    order = np.sort(np.asarray(p, np.float64))
    if order[0] >= quantile:
        return 0.0
    acc = np.float32(0.0)
    for idx in range(1, len(order)):
        acc = np.float32(np.float64(acc) + order[idx])
        if acc >= quantile:
            return float(order[idx - 1] + np.finfo(np.float32).eps)
    return -1.0       # quantile above the mass behind the smallest entry: the reference asserts(false), release builds return -1 (nothing clipped)


def apply_quantile_clipping(quantile: float, p: np.ndarray) -> np.ndarray:
    """agent.cpp:121-130: entries below get_quantile's threshold are zeroed, then renormalised."""
    thresh = get_quantile(p, quantile)
    q = np.where(p < thresh, 0.0, p)
    return q / np.cumsum(q)[-1]                 # sequential double sum, as the reference's loop adds it


class _Game:
    def __init__(self, slot: int, record: GameRecord, pos: env.Position, rng: np.random.Generator, allow_resign: bool):
        self.slot, self.record, self.pos, self.rng, self.allow_resign = slot, record, pos, rng, allow_resign
        self.samples = []               # (position clone, moves, policy, best_move_q) until the game's result is known


class SelfPlay:
    """Plays `n_games` games, `concurrent` at a time, on the trees of one SearchPool.

    start_fen(game_index) -> FEN ("" = the variant's start position); raw_policy(list of positions) -> list of probability
    vectors over each position's legal moves (needed only when mean_init_ply > 0: RawNetAgent::evaluate_board_state)."""

    def __init__(self, pool: search.SearchPool, settings: SelfPlaySettings, concurrent: int,
                 start_fen: Optional[Callable[[int], str]] = None,
                 raw_policy: Optional[Callable[[Sequence[env.Position]], List[np.ndarray]]] = None,
                 exporter=None):
        """exporter: a traindata.TrainDataExporter; every searched position becomes a training sample (generate_game,
        selfplay.cpp:237-248) -- the concurrent games buffer their samples and are written game by game as they finish."""
        self.pool, self.s, self.concurrent = pool, settings, concurrent
        self.exporter = exporter
        self.start_fen = start_fen or (lambda i: "")
        self.raw_policy = raw_policy
        self.games: List[Optional[_Game]] = [None] * concurrent
        self.finished: List[GameRecord] = []
        self.started = 0
        self.stats = dict(moves=0, nodes=0, nn_evals=0, seconds=0.0, kept_subtrees=0, restarts=0)
        for slot in range(concurrent):                      # one tree slot per concurrent game
            t = pool.add_position("", settings.is960, settings.variant)
            assert t == slot, "the pool must be empty when the game loop takes it over"

    # ---- game start: init_starting_state_from_raw_policy --------------------------------------------------------------
    def _new_game(self, slot: int) -> _Game:
        idx = self.started
        self.started += 1
        s = self.s
        rng = np.random.default_rng([s.seed, idx])
        fen = self.start_fen(idx)
        pos = env.Position(fen, s.is960, s.variant)
        rec = GameRecord(start_fen=pos.fen(), variant=("standard" if s.variant == "chess" and not s.is960 else s.variant + ("960" if s.is960 else "")),
                         white=s.white, black=s.black, event=s.event)
        if s.mean_init_ply > 0:
            if self.raw_policy is None:
                raise ValueError("mean_init_ply > 0 needs a raw_policy evaluator")
            plies = int(rng.exponential(s.mean_init_ply) + 0.5)          # random_exponential(1/mean) + 0.5, clip_ply
            plies = min(plies, s.max_init_ply)
            for _ in range(plies):
                moves = pos.legal_moves()
                if len(moves) == 0:
                    break
                p = np.ones(1) if len(moves) == 1 else np.asarray(self.raw_policy([pos])[0], np.float64)
                if rng.random() < s.raw_policy_prob_temperature:          # apply_raw_policy_temp
                    u = rng.random()
                    p = apply_temperature(p, 10.0 if u < 0.05 else 5.0 if u < 0.25 else 2.0)
                mv = moves[int(rng.choice(len(moves), p=p / p.sum()))]   # random_choice
                nxt = pos.clone()
                nxt.push(mv)
                if nxt.terminal() != env.TERMINAL_NONE:                  # leads_to_terminal: keep the game alive
                    nxt.close()
                    break
                rec.san.append(pos.move_san(mv) + " {book}")
                rec.uci.append(pos.move_uci(mv))
                pos.close()
                pos = nxt
            rec.book_plies = len(rec.uci)
        self.pool.reset_position(slot, rec.start_fen, s.is960, s.variant)
        self.pool.set_active(slot, True)
        for u in rec.uci:
            self.pool.apply_move(slot, u)
        allow_resign = s.resign_probability >= 0.01 and rng.random() < s.resign_probability
        return _Game(slot, rec, pos, rng, allow_resign)

    # ---- one move of every running game -----------------------------------------------------------------------------
    def _choose(self, g: _Game):
        """Agent::set_best_move (agent.cpp:38-55) on the root's MCTS policy."""
        s = self.s
        moves, _, _, _ = self.pool.root_children(g.slot)
        policy, best_q = self.pool.root_policy(g.slot)
        ply = len(g.record.uci)                       # steps_from_null of the game state
        if ply < s.temperature_moves and s.init_temperature > 0.01:
            p = apply_temperature(policy.copy(), s.init_temperature * s.temperature_decay ** ply)
            if s.quantile_clipping != 0:
                p = apply_quantile_clipping(s.quantile_clipping, p)
            i = int(g.rng.choice(len(p), p=p / p.sum()))
        else:
            i = int(np.argmax(policy))
        return moves[i], best_q

    def _finish(self, g: _Game, result: int, why: str):
        g.record.result, g.record.termination = result, why
        if self.exporter is not None:
            self.exporter.new_game()
            for p, moves, policy, q in g.samples:
                self.exporter.save_sample(p, moves, policy, q)
                p.close()
            self.stats["samples"] = self.stats.get("samples", 0) + self.exporter.export_game_samples(result)
            g.samples = []
        self.finished.append(g.record)
        g.pos.close()
        self.games[g.slot] = None
        # the slot's tree sits out the following pool.run calls until a new game takes it (otherwise a finished game's tree would be
        # searched to the full budget every round and its visits counted as nodes)
        self.pool.set_active(g.slot, False)

    def play(self, n_games: int, threads: int = 16) -> List[GameRecord]:
        s = self.s
        t0 = time.perf_counter()
        while len(self.finished) < n_games:
            for slot in range(self.concurrent):                # refill free slots
                if self.games[slot] is None and self.started < n_games:
                    g = self._new_game(slot)
                    self.games[slot] = g
                    self._check_over(g)                        # a start position can already be decided
            active = [g for g in self.games if g is not None]
            if not active:
                break
            budget = dict(simulations=s.simulations) if s.simulations else dict(nodes=s.nodes)
            if s.node_random_factor > 0 and s.nodes:            # adjust_node_count (one draw per round)
                span = int(s.nodes * s.node_random_factor)
                if span:
                    budget = dict(nodes=s.nodes + int(active[0].rng.integers(0, span)) - span // 2)
            st = self.pool.run(threads=threads, **budget)
            self.stats["nodes"] += st.nodes
            self.stats["nn_evals"] += st.nn_evals
            for g in active:
                mv, best_q = self._choose(g)
                if self.exporter is not None:                                  # save_sample(state, evalInfo) before the move
                    moves_all, _, _, _ = self.pool.root_children(g.slot)
                    policy_all, _ = self.pool.root_policy(g.slot)
                    g.samples.append((g.pos.clone(), moves_all, policy_all, best_q))
                uci = g.pos.move_uci(mv)
                san = g.pos.move_san(mv)
                g.pos.push(mv)
                over = self._check_over(g, san=san, uci=uci)
                self.stats["moves"] += 1
                if over:
                    continue
                if g.allow_resign and best_q < s.resign_threshold:           # check_for_resignation (after the move: side to move wins)
                    self._finish(g, 1 if g.pos.side_to_move() == 0 else -1, "resignation")
                    continue
                if len(g.record.uci) >= s.max_plies:
                    self._finish(g, 0, "ply limit")
                    continue
                if s.reuse_tree:
                    kept = self.pool.apply_move(g.slot, uci)
                else:
                    self.pool.reset_position(g.slot, g.pos.fen(), s.is960, s.variant)
                    kept = False
                self.stats["kept_subtrees" if kept else "restarts"] += 1
        self.stats["seconds"] = time.perf_counter() - t0
        return self.finished[:n_games]

    def _check_over(self, g: _Game, san: Optional[str] = None, uci: Optional[str] = None) -> bool:
        """play_move_and_update (selfplay.cpp:38-54): record the move, ask the state for the result, mark a win with '#'."""
        t = g.pos.terminal()
        if san is not None:
            if t in (env.TERMINAL_WIN, env.TERMINAL_LOSS):
                san = san[:-1] + "#" if san.endswith("+") else san + "#"
            g.record.san.append(san)
            g.record.uci.append(uci)
        if t == env.TERMINAL_NONE:
            return False
        stm_white = g.pos.side_to_move() == 0
        if t == env.TERMINAL_DRAW:
            res = 0
        elif t == env.TERMINAL_LOSS:                      # the side to move has lost
            res = -1 if stm_white else 1
        else:
            res = 1 if stm_white else -1
        self._finish(g, res, "terminal")
        return True


def net_raw_policy(net, mode: int, version_major: int, is_policy_map: bool = True):
    """raw_policy callback on a HipAPI net: RawNetAgent::evaluate_board_state (rawnetagent.cpp:45-92) -- planes, predict, gather the
    probabilities of the legal moves."""
    nbp, nbin, B = net.get_nb_policy_values(), net.get_nb_input_values_total(), net.get_batch_size()
    planes = np.zeros((B, nbin), np.float32)
    value, probs = np.zeros(B, np.float32), np.zeros(B * nbp, np.float32)

    def evaluate(positions):
        out = []
        for off in range(0, len(positions), B):
            chunk = positions[off:off + B]
            for i, p in enumerate(chunk):
                planes[i] = p.planes(mode, version_major, True).reshape(-1)
            net.predict(planes.reshape(-1), value, probs)
            pr = probs.reshape(B, nbp)
            for i, p in enumerate(chunk):
                out.append(np.array([pr[i, p.policy_index(m, mode, is_policy_map)] for m in p.legal_moves()], np.float64))
        return out

    return evaluate


@dataclass
class TournamentResult:
    """TournamentResult of go_arena (selfplay.cpp:387-424): seen from the contender (player A)."""
    player_a: str = "A"
    player_b: str = "B"
    wins: int = 0
    draws: int = 0
    losses: int = 0

    def score(self) -> float:
        n = self.wins + self.draws + self.losses
        return (self.wins + 0.5 * self.draws) / n if n else 0.0


class Arena:
    """Two players (search pools with their own nets) play `n_games` against each other, `concurrent` games at a time.

    go_arena (selfplay.cpp:387-424): game 2i has the contender A as White from a fresh start position, game 2i+1 replays the SAME
    start position with colours swapped.  generate_arena_game (:267-308): the player to move searches, both players apply the move
    to their trees (own move / opponent's move), always the best move (no temperature), no resignation.  Each game owns tree slot g
    in BOTH pools; the pool of the player that is not to move pauses that tree (mi_search_set_active)."""

    def __init__(self, pool_a: search.SearchPool, pool_b: search.SearchPool, settings: SelfPlaySettings, concurrent: int,
                 start_fen: Optional[Callable[[int], str]] = None, names=("contender", "champion")):
        self.pools, self.s, self.concurrent, self.names = (pool_a, pool_b), settings, concurrent, names
        self.start_fen = start_fen or (lambda i: "")
        for slot in range(concurrent):
            for p in self.pools:
                assert p.add_position("", settings.is960, settings.variant) == slot
        self.stats = dict(moves=0, nodes=0, seconds=0.0)

    def play(self, n_games: int, threads: int = 16):
        s = self.s
        res = TournamentResult(self.names[0], self.names[1])
        games: List[Optional[dict]] = [None] * self.concurrent
        records: List[GameRecord] = []
        started, pair_fen = 0, {}
        t0 = time.perf_counter()
        while len(records) < n_games:
            for slot in range(self.concurrent):
                if games[slot] is None and started < n_games:
                    idx = started
                    started += 1
                    if idx % 2 == 0:
                        pos = env.Position(self.start_fen(idx // 2), s.is960, s.variant)
                        pair_fen[idx // 2] = pos.fen()
                    else:
                        pos = env.Position(pair_fen[idx // 2], s.is960, s.variant)       # gamePGN.fen of the game before
                    a_white = idx % 2 == 0
                    rec = GameRecord(start_fen=pos.fen(), variant=s.variant + ("960" if s.is960 else ""), event="Arena",
                                     white=self.names[0 if a_white else 1], black=self.names[1 if a_white else 0])
                    for p in self.pools:
                        p.reset_position(slot, rec.start_fen, s.is960, s.variant)
                    games[slot] = dict(idx=idx, pos=pos, rec=rec, a_white=a_white)
            active = [(slot, g) for slot, g in enumerate(games) if g is not None]
            if not active:
                break
            mover = {}
            for slot, g in active:                       # which player searches this game now
                white_to_move = g["pos"].side_to_move() == 0
                mover[slot] = 0 if white_to_move == g["a_white"] else 1
            for pi, p in enumerate(self.pools):
                for slot in range(self.concurrent):
                    p.set_active(slot, games[slot] is not None and mover.get(slot) == pi)
                if any(mover[slot] == pi for slot, _ in active):
                    st = p.run(simulations=s.simulations, nodes=s.nodes, threads=threads)
                    self.stats["nodes"] += st.nodes
            for slot, g in active:
                p = self.pools[mover[slot]]
                moves, _, _, _ = p.root_children(slot)
                policy, _ = p.root_policy(slot)
                mv = moves[int(np.argmax(policy))]
                uci, san = g["pos"].move_uci(mv), g["pos"].move_san(mv)
                g["pos"].push(mv)
                t = g["pos"].terminal()
                if t in (env.TERMINAL_WIN, env.TERMINAL_LOSS):
                    san = san[:-1] + "#" if san.endswith("+") else san + "#"
                g["rec"].san.append(san)
                g["rec"].uci.append(uci)
                self.stats["moves"] += 1
                over = t != env.TERMINAL_NONE or len(g["rec"].uci) >= s.max_plies
                if not over:
                    for q in self.pools:                 # own move in one tree, the opponent's move in the other
                        q.apply_move(slot, uci)
                    continue
                stm_white = g["pos"].side_to_move() == 0
                result = 0 if t in (env.TERMINAL_DRAW, env.TERMINAL_NONE) else ((-1 if stm_white else 1) if t == env.TERMINAL_LOSS else (1 if stm_white else -1))
                g["rec"].result, g["rec"].termination = result, "terminal" if t != env.TERMINAL_NONE else "ply limit"
                a_score = result if g["a_white"] else -result
                if a_score > 0:
                    res.wins += 1
                elif a_score < 0:
                    res.losses += 1
                else:
                    res.draws += 1
                records.append(g["rec"])
                g["pos"].close()
                games[slot] = None
        self.stats["seconds"] = time.perf_counter() - t0
        return res, records
