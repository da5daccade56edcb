(* hnsw_mi355x.ml -- OCaml side of the drop-in: ctypes-foreign binding of libhnsw_mi355x.so
   (include/hnsw_mi355x.h) plus the flatten shims that turn the reference's own graph containers
   into the tables the C ABI takes.

   SOURCE ONLY in this repository: the build image has no OCaml toolchain (no ocaml / opam / dune),
   so this file is not compiled or tested here; the identical ABI is exercised from Python ctypes
   (tests/) and C++ (host/hnsw_front.hpp).  It is written against the reference's modules as they
   are: Ohnsw (lib/ohnsw.ml) and Hnsw.Ba (lib/hnsw.ml:817-819).

   What stays OCaml: the graph builder (Ohnsw.build_batch_bigarray, Hnsw.Ba.build) and every
   signature.  What moves: the bodies of knn / knn_batch*, i.e. the search_one + search_k loops. *)
open Ctypes
open Foreign

let lib = Dl.dlopen ~filename:"libhnsw_mi355x.so" ~flags:[ Dl.RTLD_NOW ]

(* ---- C structs ---------------------------------------------------------------------------- *)
type layer_desc
let layer_desc : layer_desc structure typ = structure "hnsw_layer_desc"
let ld_n_nodes = field layer_desc "n_nodes" int64_t
let ld_nodes = field layer_desc "nodes" (ptr int64_t)
let ld_deg = field layer_desc "deg" (ptr int32_t)
let ld_nbr = field layer_desc "nbr" (ptr int32_t)
let () = seal layer_desc

type index_desc
let index_desc : index_desc structure typ = structure "hnsw_index_desc"
let d_vectors = field index_desc "vectors" (ptr float)
let d_n = field index_desc "n" int64_t
let d_d = field index_desc "d" int32_t
let d_row_stride = field index_desc "row_stride" int64_t
let d_metric = field index_desc "metric" int32_t
let d_id_base = field index_desc "id_base" int32_t
let d_max_degree0 = field index_desc "max_degree0" int32_t
let d_max_degree = field index_desc "max_degree" int32_t
let d_max_layer = field index_desc "max_layer" int32_t
let d_entry_point = field index_desc "entry_point" int64_t
let d_deg0 = field index_desc "deg0" (ptr int32_t)
let d_nbr0 = field index_desc "nbr0" (ptr int32_t)
let d_upper = field index_desc "upper" (ptr layer_desc)
let () = seal index_desc

type search_params
let search_params : search_params structure typ = structure "hnsw_search_params"
let p_ef = field search_params "ef" int32_t
let p_k = field search_params "k" int32_t
let p_fill = field search_params "fill" int32_t
let p_semantics = field search_params "semantics" int32_t
let () = seal search_params

type index = unit ptr
let index : index typ = ptr void

(* ---- entry points (the OCaml 4.x runtime lock is released around the blocking calls) -------- *)
let hnsw_last_error = foreign ~from:lib "hnsw_last_error" (void @-> returning string)
let hnsw_index_create =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_create"
    (ptr index_desc @-> int32_t @-> ptr index @-> returning int32_t)
let hnsw_index_destroy = foreign ~from:lib "hnsw_index_destroy" (index @-> returning int32_t)
let hnsw_search_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_batch"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr int32_t @-> ptr float
     @-> ptr uint32_t @-> ptr uint32_t @-> returning int32_t)
let hnsw_search_layer_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_layer_batch"
    (index @-> int32_t @-> ptr float @-> int64_t @-> int64_t @-> ptr int64_t @-> int32_t
     @-> ptr search_params @-> ptr int32_t @-> ptr float @-> ptr int32_t @-> ptr uint32_t
     @-> ptr uint32_t @-> returning int32_t)
let hnsw_search_one_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_one_batch"
    (index @-> int32_t @-> ptr float @-> int64_t @-> int64_t @-> ptr int64_t @-> ptr int64_t
     @-> ptr float @-> returning int32_t)
let hnsw_index_set_option =
  foreign ~from:lib "hnsw_index_set_option" (index @-> string @-> int64_t @-> returning int32_t)
let hnsw_index_kernel_times =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_kernel_times"
    (index @-> ptr double @-> ptr double @-> ptr int32_t @-> returning int32_t)
type request = unit ptr
let request : request typ = ptr void
let hnsw_search_submit =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_submit"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr request @-> returning int32_t)
let hnsw_search_wait =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_wait"
    (request @-> ptr int32_t @-> ptr float @-> ptr uint32_t @-> ptr uint32_t @-> returning int32_t)
type multi = unit ptr
let multi : multi typ = ptr void
let hnsw_multi_create =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_multi_create"
    (ptr index_desc @-> ptr int32_t @-> int32_t @-> ptr multi @-> returning int32_t)
let hnsw_multi_destroy = foreign ~from:lib "hnsw_multi_destroy" (multi @-> returning int32_t)
let hnsw_multi_search_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_multi_search_batch"
    (multi @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr int32_t @-> ptr float
     @-> ptr uint32_t @-> ptr uint32_t @-> returning int32_t)

(* Error convention -> the reference's exceptions (lib/ohnsw.ml:25,343,862) *)
let check rc =
  match Int32.to_int rc with
  | 0 -> ()
  | -1 | -2 | -3 -> invalid_arg (hnsw_last_error ())   (* BAD_ARG | EMPTY_INDEX | DEGREE_OVERFLOW *)
  | _ -> failwith (hnsw_last_error ())

(* ---- flatten ------------------------------------------------------------------------------- *)
module A1 = Bigarray.Array1
module A2 = Bigarray.Array2

type flat = {
  deg0 : (int32, Bigarray.int32_elt, Bigarray.c_layout) A1.t;
  nbr0 : (int32, Bigarray.int32_elt, Bigarray.c_layout) A2.t;   (* n x 2M, iteration order *)
  upper : ((int64, Bigarray.int64_elt, Bigarray.c_layout) A1.t
           * (int32, Bigarray.int32_elt, Bigarray.c_layout) A1.t
           * (int32, Bigarray.int32_elt, Bigarray.c_layout) A2.t) array;
  entry_point : int;
  max_layer : int;
}

(* Ohnsw.Hgraph.t (lib/ohnsw.ml:307-312): every layer is dense over all n nodes (:328-330);
   Graph.iter_neighbours (:176-180) walks nodes, Neighbours.iter (:127) walks a list head first --
   exactly the order search_k folds over (:570).  A list longer than its row raises: the C side
   would refuse it too (HNSW_ERR_DEGREE_OVERFLOW); nothing is ever truncated. *)
let flatten_ohnsw (h : _ Ohnsw.Hgraph.t) ~num_connections:m : flat =
  let n = Ohnsw.Hgraph.num_nodes h in
  let max_layer = Ohnsw.Hgraph.max_layer h in
  let row g node width (dst : (int32, _, _) A2.t) r =
    let nb = Ohnsw.Graph.adjacent g node in
    if Ohnsw.Neighbours.length nb > width then invalid_arg "flatten: degree exceeds row width";
    let j = ref 0 in
    Ohnsw.Neighbours.iter nb ~f:(fun e -> dst.{r, !j} <- Int32.of_int e; incr j);
    !j
  in
  let deg0 = A1.create Bigarray.int32 Bigarray.c_layout n in
  let nbr0 = A2.create Bigarray.int32 Bigarray.c_layout n (2 * m) in
  A2.fill nbr0 (-1l);
  let g0 = Ohnsw.Hgraph.layer h 0 in
  for i = 0 to n - 1 do deg0.{i} <- Int32.of_int (row g0 i (2 * m) nbr0 i) done;
  let upper =
    Array.init max_layer (fun l ->
        let g = Ohnsw.Hgraph.layer h (l + 1) in
        (* nodes present on the layer: those with links there, plus the entry point *)
        let present i =
          Ohnsw.Neighbours.length (Ohnsw.Graph.adjacent g i) > 0
          || Ohnsw.Hgraph.entry_point h = Some i in
        let ids = List.filter present (List.init n (fun i -> i)) in
        let c = List.length ids in
        let nodes = A1.create Bigarray.int64 Bigarray.c_layout c in
        let deg = A1.create Bigarray.int32 Bigarray.c_layout c in
        let nbr = A2.create Bigarray.int32 Bigarray.c_layout c m in
        A2.fill nbr (-1l);
        List.iteri (fun s i ->
            nodes.{s} <- Int64.of_int i;
            deg.{s} <- Int32.of_int (row g i m nbr s)) ids;
        (nodes, deg, nbr))
  in
  { deg0; nbr0; upper; max_layer;
    entry_point = (match Ohnsw.Hgraph.entry_point h with Some e -> e | None -> -1) }

(* Hnsw.Ba.Hgraph.t (lib/hnsw.ml:342-348): layers = Map int -> MapGraph.t, a MapGraph holding
   connections = Map int -> NeighbourList.t ({list; length}, lib/hnsw.ml:33-45).  Node ids are the
   1-based matrix columns (lib/hnsw.ml:325); a node without an entry in a layer's map has no
   neighbours there (MapGraph.adjacent, lib/hnsw.ml:146-149).  NeighbourList.fold walks the list head
   first (lib/hnsw.ml:44-45), which is the order Search.search folds over (lib/hnsw_algo.ml:381-383).
   Pass the result to [create ~id_base:1]. *)
let flatten_ba (h : Hnsw.Ba.Hgraph.t) ~num_connections:m : flat =
  let module G = Hnsw.Ba.Hgraph in
  let n = G.max_node_id h in                       (* = Values.length, lib/hnsw.ml:394 *)
  let max_layer = G.max_layer h in
  let fill_row (nl : Hnsw.NeighbourList.t) width (dst : (int32, _, _) A2.t) r =
    if Hnsw.NeighbourList.length nl > width then invalid_arg "flatten: degree exceeds row width";
    let j = ref 0 in
    Hnsw.NeighbourList.fold nl ~init:() ~f:(fun () e -> dst.{r, !j} <- Int32.of_int e; incr j);
    !j
  in
  let deg0 = A1.create Bigarray.int32 Bigarray.c_layout n in
  let nbr0 = A2.create Bigarray.int32 Bigarray.c_layout n (2 * m) in
  A1.fill deg0 0l; A2.fill nbr0 (-1l);
  let g0 = G.layer h 0 in
  Base.Map.iteri g0.Hnsw.MapGraph.connections ~f:(fun ~key ~data ->
      deg0.{key - 1} <- Int32.of_int (fill_row data (2 * m) nbr0 (key - 1)));
  let upper =
    Array.init max_layer (fun l ->
        let g = G.layer h (l + 1) in
        let c = Base.Map.length g.Hnsw.MapGraph.connections in
        let nodes = A1.create Bigarray.int64 Bigarray.c_layout c in
        let deg = A1.create Bigarray.int32 Bigarray.c_layout c in
        let nbr = A2.create Bigarray.int32 Bigarray.c_layout c m in
        A2.fill nbr (-1l);
        let s = ref 0 in
        Base.Map.iteri g.Hnsw.MapGraph.connections ~f:(fun ~key ~data ->
            nodes.{!s} <- Int64.of_int key;
            deg.{!s} <- Int32.of_int (fill_row data m nbr !s);
            incr s);
        (nodes, deg, nbr))
  in
  { deg0; nbr0; upper; max_layer; entry_point = G.entry_point h (* -1 when empty, lib/hnsw.ml:391 *) }

(* The inverse: rebuild an Ohnsw.Hgraph.t from the flattened tables (e.g. an index built on the device by
   hnsw_build and fetched with hnsw_index_export_*), so that the OCaml builder can keep inserting into
   it.  Every Neighbours.t gets its list in the stored order (Neighbours.iter order = search_k's
   fold order, :127,:570): Neighbours.add conses (:116-118), so a row is pushed back to front.  The
   tables hold symmetric links already, so the rows are written with Vector.set, not with
   Graph.set_connections (which would add every link twice, :182-196). *)
let unflatten_ohnsw (distance : 'a Ohnsw.distance) (value : 'a Ohnsw.value) (f : flat) : 'a Ohnsw.Hgraph.t =
  let h = Ohnsw.Hgraph.create distance value in
  let n = A1.dim f.deg0 in
  for _ = 1 to n do ignore (Ohnsw.Hgraph.add_node h) done;          (* lib/ohnsw.ml:328-330 *)
  Ohnsw.Hgraph.set_max_layer h f.max_layer;                            (* :347-351 *)
  let neighbours_of_row (nbr : (int32, _, _) A2.t) r deg =
    let nb = Ohnsw.Neighbours.create () in
    for j = deg - 1 downto 0 do Ohnsw.Neighbours.add nb (Int32.to_int nbr.{r, j}) done;
    nb in
  let g0 = Ohnsw.Hgraph.layer h 0 in
  for i = 0 to n - 1 do
    Ohnsw.Vector.set g0 i (neighbours_of_row f.nbr0 i (Int32.to_int f.deg0.{i}))
  done;
  Array.iteri (fun l (nodes, deg, nbr) ->
      let g = Ohnsw.Hgraph.layer h (l + 1) in
      for r = 0 to A1.dim nodes - 1 do
        Ohnsw.Vector.set g (Int64.to_int nodes.{r}) (neighbours_of_row nbr r (Int32.to_int deg.{r}))
      done) f.upper;
  if f.entry_point >= 0 then Ohnsw.Hgraph.set_entry_point h f.entry_point;   (* :341-344 *)
  h

(* ---- device-resident index ------------------------------------------------------------------ *)
type t = { handle : index; k_base : int; dim : int }

let create ?(device = 0) ?(metric = 0) ~id_base ~num_connections:m (vectors : Lacaml.S.mat) (f : flat) : t =
  (* a Lacaml.S.mat is a Fortran-layout dim x n Bigarray: in memory, n rows of dim floats *)
  let dim = A2.dim1 vectors and n = A2.dim2 vectors in
  let layers = CArray.make layer_desc (max 1 f.max_layer) in
  Array.iteri (fun l (nodes, deg, nbr) ->
      let ld = CArray.get layers l in
      setf ld ld_n_nodes (Int64.of_int (A1.dim nodes));
      setf ld ld_nodes (bigarray_start array1 nodes);
      setf ld ld_deg (bigarray_start array1 deg);
      setf ld ld_nbr (bigarray_start array2 nbr)) f.upper;
  let d = make index_desc in
  setf d d_vectors (bigarray_start array2 vectors);
  setf d d_n (Int64.of_int n); setf d d_d (Int32.of_int dim);
  setf d d_row_stride (Int64.of_int dim);
  setf d d_metric (Int32.of_int metric); setf d d_id_base (Int32.of_int id_base);
  setf d d_max_degree0 (Int32.of_int (2 * m)); setf d d_max_degree (Int32.of_int m);
  setf d d_max_layer (Int32.of_int f.max_layer);
  setf d d_entry_point (Int64.of_int f.entry_point);
  setf d d_deg0 (bigarray_start array1 f.deg0); setf d d_nbr0 (bigarray_start array2 f.nbr0);
  setf d d_upper (CArray.start layers);
  let out = allocate index null in
  check (hnsw_index_create (addr d) (Int32.of_int device) out);
  let t = { handle = !@out; k_base = id_base; dim } in
  Gc.finalise (fun t -> ignore (hnsw_index_destroy t.handle)) t;
  t

let search ?(semantics = 0) t (batch : Lacaml.S.mat) ~ef ~k ~fill =
  let nq = A2.dim2 batch in
  (* results as the reference lays them out: k x nq Fortran = [nq][k] in memory *)
  let distances = Lacaml.S.Mat.create k nq in
  let ids = A2.create Bigarray.int32 Bigarray.fortran_layout k nq in
  let p = make search_params in
  setf p p_ef (Int32.of_int ef); setf p p_k (Int32.of_int k); setf p p_fill (Int32.of_int fill);
  setf p p_semantics (Int32.of_int semantics);
  check (hnsw_search_batch t.handle (bigarray_start array2 batch) (Int64.of_int nq)
           (Int64.of_int t.dim) (addr p) (bigarray_start array2 ids)
           (bigarray_start array2 distances) (from_voidp uint32_t null) (from_voidp uint32_t null));
  ids, distances

(* ---- the drop-in bodies ---------------------------------------------------------------------- *)

(* Ohnsw.knn_batch_bigarray : 'a Hgraph.t -> k:int -> Lacaml.S.mat -> int array array * Lacaml.S.mat
   (lib/ohnsw.ml:877-897): same signature once the index handle is cached next to the hgraph. *)
let ohnsw_knn_batch_bigarray (t : t) ~k (batch : Lacaml.S.mat) =
  let ids32, distances = search t batch ~ef:k ~k ~fill:0 (* NaN / -1, :880-881 *) in
  let nq = A2.dim2 batch in
  let ids = Array.init nq (fun j -> Array.init k (fun i -> Int32.to_int ids32.{i + 1, j + 1})) in
  ids, distances

(* Hnsw.Ba.knn_batch : t -> Lacaml.S.mat -> num_neighbours_search:int -> num_neighbours:int
   -> Lacaml.S.mat (lib/hnsw.ml:769-777): distances only, +inf filled (:771). *)
let ba_knn_batch ?(nearest_k_compat = false) (t : t) (batch : Lacaml.S.mat) ~num_neighbours_search ~num_neighbours =
  (* semantics 1 = Nearest.insert_distance rule; 2 = the same + Nearest.nearest_k's output (the k
     farthest of W when num_neighbours_search > num_neighbours, lib/hnsw.ml:522-525) for callers
     that need the reference's result bit for bit *)
  snd (search ~semantics:(if nearest_k_compat then 2 else 1) t batch ~ef:num_neighbours_search ~k:num_neighbours ~fill:1)

(* Ohnsw.search_k (lib/ohnsw.ml:543-588) on one layer for ONE target: the start MinQueue as a node
   list, the result MinQueue as an ascending (node, distance) list.  ~semantics:1 is
   Hnsw_algo.Search.search (lib/hnsw_algo.ml:350-391). *)
let search_k ?(semantics = 0) (t : t) ~layer ~(start_nodes : int list) (target : Lacaml.S.vec) ~k =
  let ns = List.length start_nodes in
  let st = CArray.of_list int64_t (List.map Int64.of_int start_nodes) in
  let ids = CArray.make int32_t k and dist = CArray.make float k in
  let cnt = allocate int32_t 0l in
  let p = make search_params in
  setf p p_ef (Int32.of_int k); setf p p_k (Int32.of_int k); setf p p_fill 0l;
  setf p p_semantics (Int32.of_int semantics);
  check (hnsw_search_layer_batch t.handle (Int32.of_int layer) (bigarray_start array1 target) 1L
           (Int64.of_int t.dim) (CArray.start st) (Int32.of_int ns) (addr p) (CArray.start ids)
           (CArray.start dist) cnt (from_voidp uint32_t null) (from_voidp uint32_t null));
  List.init (Int32.to_int !@cnt) (fun i -> Int32.to_int (CArray.get ids i), CArray.get dist i)

(* Ohnsw.search_one (lib/ohnsw.ml:492-512): the node the greedy walk on `layer` ends on *)
let search_one (t : t) ~layer ~start_node (target : Lacaml.S.vec) =
  let st = allocate int64_t (Int64.of_int start_node) and node = allocate int64_t 0L in
  check (hnsw_search_one_batch t.handle (Int32.of_int layer) (bigarray_start array1 target) 1L
           (Int64.of_int t.dim) st node (from_voidp float null));
  Int64.to_int !@node

(* Batches in flight: [submit] copies the batch in and starts the search, [wait] returns what
   [search] would have.  A caller with more than one batch overlaps them
   (let r1 = submit t b1 ... in let r2 = submit t b2 ... in wait r1; wait r2): the next batch fills
   the drain of the previous one (12.9 M q/s against 9.4 M q/s for back-to-back synchronous calls on
   the SIFT1M-shaped workload, host buffers included). *)
type pending = { req : request; p_k : int; p_nq : int; keep : Lacaml.S.mat }

let submit ?(semantics = 0) t (batch : Lacaml.S.mat) ~ef ~k ~fill : pending =
  let nq = A2.dim2 batch in
  let p = make search_params in
  setf p p_ef (Int32.of_int ef); setf p p_k (Int32.of_int k); setf p p_fill (Int32.of_int fill);
  setf p p_semantics (Int32.of_int semantics);
  let out = allocate request null in
  check (hnsw_search_submit t.handle (bigarray_start array2 batch) (Int64.of_int nq)
           (Int64.of_int t.dim) (addr p) out);
  { req = !@out; p_k = k; p_nq = nq; keep = batch }

let wait (r : pending) =
  let distances = Lacaml.S.Mat.create r.p_k r.p_nq in
  let ids = A2.create Bigarray.int32 Bigarray.fortran_layout r.p_k r.p_nq in
  check (hnsw_search_wait r.req (bigarray_start array2 ids) (bigarray_start array2 distances)
           (from_voidp uint32_t null) (from_voidp uint32_t null));
  ids, distances
