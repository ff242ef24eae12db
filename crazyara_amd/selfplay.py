"""Self-play / arena game loop on top of the search pool.

Follows engine/src/rl/selfplay.cpp (`generate_game` :192-265, `generate_arena_game` :267-308, `go_arena` :387-424,
`init_starting_state_from_raw_policy` :426-452, `play_move_and_update` :38-54), agents/agent.cpp (`set_best_move` :38-55) and
rl/gamepgn.cpp (:28-56) -- restructured for the many-trees pool: G games are played CONCURRENTLY, every game owns one tree
slot, one `run` of the pool searches the next move of all of them in shared GPU batches, then every game picks and plays its
move (the searched subtree is kept: mi_search_apply_move) and finished games are replaced by new ones.

The loops themselves are native (csrc/rl/selfplay.cpp behind mi_selfplay_*, include/crazyara_hip.h): this module holds the settings,
the game records with the reference's PGN dialect, and thin handles that start a loop inside the library and read the finished games
back.  Where the reference draws from rand() / std::random_device (opening plies, temperature sampling, resignation, node-count
jitter) a game draws from its own seeded generator, so a run replays; the distributions are the reference's.  apply_temperature /
get_quantile / apply_quantile_clipping below restate the library's helpers in numpy for the tests that compare them with the
reference's compiled functions.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _capi, env, search

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
    quick_search_probability: float = 0.0   # RLSettings::quickSearchProbability: a move searched quickly is not exported (:154-159,213-224)
    quick_search_nodes: int = 100           # RLSettings::quickSearchNodes
    quick_search_q_value_weight: float = 0.7    # RLSettings::quickSearchQValueWeight
    quick_dirichlet_epsilon: float = 0.0    # RLSettings::quickDirichletEpsilon
    low_policy_clip_threshold: float = 0.0  # RLSettings::lowPolicyClipThreshold: sharpen_distribution on the exported policy (:229-231)
    num_phases: int = 1                 # MCTSAgent::get_num_phases: > 1 = one exporter per game phase (:232-238)
    game_phase_definition: int = 0      # SearchSettings::gamePhaseDefinition: 0 lichess, 1 movecount (board.cpp:540-587)
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


class SelfPlaySettingsC(C.Structure):
    """mi_selfplay_settings (include/crazyara_hip.h)"""
    _fields_ = [("simulations", C.c_uint), ("nodes", C.c_uint), ("node_random_factor", C.c_float), ("mean_init_ply", C.c_float),
                ("max_init_ply", C.c_int), ("raw_policy_prob_temperature", C.c_float), ("init_temperature", C.c_float),
                ("temperature_moves", C.c_int), ("temperature_decay", C.c_float), ("quantile_clipping", C.c_float),
                ("resign_probability", C.c_float), ("resign_threshold", C.c_float), ("reuse_tree", C.c_int), ("max_plies", C.c_int),
                ("seed", C.c_ulonglong), ("quick_search_probability", C.c_float), ("quick_search_nodes", C.c_uint),
                ("quick_search_q_value_weight", C.c_float), ("quick_dirichlet_epsilon", C.c_float), ("low_policy_clip_threshold", C.c_float),
                ("num_phases", C.c_int), ("game_phase_definition", C.c_int)]


class SelfPlayStatsC(C.Structure):
    """mi_selfplay_stats"""
    _fields_ = [("moves", C.c_ulonglong), ("nodes", C.c_ulonglong), ("nn_evals", C.c_ulonglong), ("kept_subtrees", C.c_ulonglong),
                ("restarts", C.c_ulonglong), ("samples", C.c_ulonglong), ("seconds", C.c_double), ("wins", C.c_int), ("draws", C.c_int),
                ("losses", C.c_int), ("reserved", C.c_int), ("run_seconds", C.c_double), ("move_seconds", C.c_double),
                ("quick_searches", C.c_ulonglong), ("samples_dropped", C.c_ulonglong)]


def _settings_c(lib, s: SelfPlaySettings) -> SelfPlaySettingsC:
    c = SelfPlaySettingsC()
    lib.mi_selfplay_default_settings(C.byref(c))
    for name, _ in SelfPlaySettingsC._fields_:
        setattr(c, name, type(getattr(c, name))(getattr(s, name)))
    return c


class _NativeLoop:
    """Handle of a native game loop (csrc/rl/selfplay.cpp behind mi_selfplay_*): the games are played inside the library; Python only
    reads the finished games back."""

    def __init__(self, pool_a, pool_b, settings: SelfPlaySettings, concurrent: int, start_fen, exporter=None, n_fens: int = 1024):
        self._lib = _capi.load()
        self.s, self.concurrent = settings, concurrent
        c = _settings_c(self._lib, settings)
        self._h = self._lib.mi_selfplay_create(pool_a._h, pool_b._h if pool_b is not None else None, C.byref(c), int(concurrent),
                                               settings.variant.encode(), int(settings.is960), exporter._h if exporter is not None else None)
        if not self._h:
            raise RuntimeError(_capi.last_error())
        self._pools = (pool_a, pool_b, exporter)           # keep them alive as long as the loop
        self._start_fen, self._n_fens_sent = start_fen, 0
        self._read = 0

    def set_epd_file(self, path: str) -> None:
        """RLSettings.epdFilePath / UCI EPD_File_Path: every game (arena: every pair) starts from a random line of the file
        (load_random_fen, rl/selfplay.cpp:58-80); "" or "<empty>" switches it off."""
        if self._lib.mi_selfplay_set_epd_file(self._h, (path or "").encode()):
            raise (ValueError if "EPD" in _capi.last_error() else RuntimeError)(_capi.last_error())

    def _send_fens(self, n: int):
        if self._start_fen is None or n <= self._n_fens_sent:
            return
        fens = [self._start_fen(i) or "" for i in range(n)]
        if self._lib.mi_selfplay_set_start_fens(self._h, ("\n".join(fens) + "\n").encode()):
            raise RuntimeError(_capi.last_error())
        self._n_fens_sent = n

    def _run(self, n_games: int, threads: int) -> int:
        n = self._lib.mi_selfplay_play(self._h, int(n_games), int(threads))
        if n < 0:
            raise RuntimeError(_capi.last_error())
        return n

    def _game(self, i: int):
        res, book, cw = C.c_int(), C.c_int(), C.c_int()
        n = self._lib.mi_selfplay_game(self._h, i, C.byref(res), C.byref(book), C.byref(cw), None, 0)
        if n < 0:
            raise RuntimeError(_capi.last_error())
        buf = C.create_string_buffer(n + 1)
        if self._lib.mi_selfplay_game(self._h, i, None, None, None, buf, n + 1) < 0:
            raise RuntimeError(_capi.last_error())
        fen, why, san, uci = buf.value.decode().split("\n")[:4]
        return fen, why, [m for m in san.split("\t") if m], [m for m in uci.split("\t") if m], res.value, book.value, bool(cw.value)

    def _stats(self) -> SelfPlayStatsC:
        st = SelfPlayStatsC()
        if self._lib.mi_selfplay_get_stats(self._h, C.byref(st)):
            raise RuntimeError(_capi.last_error())
        return st

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi_selfplay_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def _variant_tag(s: SelfPlaySettings) -> str:
    return "standard" if s.variant == "chess" and not s.is960 else s.variant + ("960" if s.is960 else "")


class SelfPlay(_NativeLoop):
    """Plays `n_games` games, `concurrent` at a time, on the trees of one SearchPool -- inside the library (csrc/rl/selfplay.cpp).

    start_fen(game_index) -> FEN ("" = the variant's start position).  The raw policy of the opening plies (mean_init_ply > 0) is read
    from the pool's own evaluator: the root priors of a freshly reset tree are RawNetAgent::evaluate_board_state's policy over the legal
    moves; `raw_policy` is accepted for compatibility with the earlier Python loop and not used.
    exporter: a traindata.TrainDataExporter; every searched position becomes a training sample (generate_game, selfplay.cpp:237-248) --
    the concurrent games buffer their samples and are written game by game as they finish."""

    def __init__(self, pool: search.SearchPool, settings: SelfPlaySettings, concurrent: int,
                 start_fen: Optional[Callable[[int], str]] = None,
                 raw_policy: Optional[Callable[[Sequence[env.Position]], List[np.ndarray]]] = None,
                 exporter=None):
        exporters = list(exporter) if isinstance(exporter, (list, tuple)) else [exporter]       # one per game phase (num_phases > 1)
        exporter = exporters[0]
        super().__init__(pool, None, settings, concurrent, start_fen, exporter)
        for phase, e in enumerate(exporters[1:], start=1):
            if self._lib.mi_selfplay_set_phase_exporter(self._h, phase, e._h):
                raise ValueError(_capi.last_error())
        self._phase_exporters = exporters
        self.pool, self.exporter = pool, exporter
        self.finished: List[GameRecord] = []
        self.stats = dict(moves=0, nodes=0, nn_evals=0, seconds=0.0, kept_subtrees=0, restarts=0)

    def play(self, n_games: int, threads: int = 16) -> List[GameRecord]:
        self._send_fens(n_games or 1024)
        total = self._run(n_games, threads)
        s = self.s
        while self._read < total:
            fen, why, san, uci, result, book, _ = self._game(self._read)
            self._read += 1
            self.finished.append(GameRecord(start_fen=fen, variant=_variant_tag(s), san=san, uci=uci, book_plies=book, result=result,
                                            termination=why, white=s.white, black=s.black, event=s.event))
        st = self._stats()
        self.stats.update(moves=st.moves, nodes=st.nodes, nn_evals=st.nn_evals, seconds=st.seconds, kept_subtrees=st.kept_subtrees,
                          restarts=st.restarts, run_seconds=st.run_seconds, move_seconds=st.move_seconds)
        self.stats.update(quick_searches=st.quick_searches, samples_dropped=st.samples_dropped)
        if self.exporter is not None:
            self.stats["samples"] = st.samples
        return self.finished[:n_games] if n_games else self.finished


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


class Arena(_NativeLoop):
    """Two players (search pools with their own nets) play `n_games` against each other, `concurrent` games at a time, inside the library.

    go_arena (selfplay.cpp:387-424): game 2i has the contender A as White from a fresh start position, game 2i+1 replays the SAME
    start position with colours swapped.  generate_arena_game (:267-308): the player to move searches, both players apply the move
    to their trees (own move / opponent's move), always the best move (no temperature), no resignation.  Each game owns tree slot g
    in BOTH pools; the pool of the player that is not to move pauses that tree.  start_fen(pair_index) -> FEN."""

    def __init__(self, pool_a: search.SearchPool, pool_b: search.SearchPool, settings: SelfPlaySettings, concurrent: int,
                 start_fen: Optional[Callable[[int], str]] = None, names=("contender", "champion")):
        super().__init__(pool_a, pool_b, settings, concurrent, start_fen)
        self.pools, self.names = (pool_a, pool_b), names
        self.stats = dict(moves=0, nodes=0, seconds=0.0, run_seconds=0.0, move_seconds=0.0)
        self.records: List[GameRecord] = []

    def play(self, n_games: int, threads: int = 16):
        self._send_fens((n_games + 1) // 2)
        total = self._run(n_games, threads)
        s = self.s
        while self._read < total:
            fen, why, san, uci, result, _, a_white = self._game(self._read)
            self._read += 1
            self.records.append(GameRecord(start_fen=fen, variant=s.variant + ("960" if s.is960 else ""), san=san, uci=uci, result=result,
                                           termination=why, event="Arena", white=self.names[0 if a_white else 1],
                                           black=self.names[1 if a_white else 0]))
        st = self._stats()
        self.stats.update(moves=st.moves, nodes=st.nodes, seconds=st.seconds, run_seconds=st.run_seconds, move_seconds=st.move_seconds)
        return TournamentResult(self.names[0], self.names[1], st.wins, st.draws, st.losses), self.records[:n_games]
